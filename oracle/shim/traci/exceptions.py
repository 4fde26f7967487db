class FatalTraCIError(Exception):
    pass


class TraCIException(Exception):
    pass
