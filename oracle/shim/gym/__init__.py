"""TEST INFRASTRUCTURE — container-only stand-in for `gym` so /root/reference/endtoend.py imports."""
from . import spaces, utils  # noqa: F401


class Env(object):
    pass
