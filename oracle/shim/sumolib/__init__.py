"""TEST INFRASTRUCTURE — import-time stand-in for SUMO's `sumolib` (never called)."""


def checkBinary(name):
    return name
