"""TEST INFRASTRUCTURE — utils/policy.py:37, 42 construct Adam optimizers that the inference path never steps
(see keras/__init__.py)."""


class Adam(object):
    def __init__(self, *a, name=None, **k):
        self._name = name
