"""TEST INFRASTRUCTURE — utils/policy.py:11 constructs PolynomialDecay schedules for optimizers the inference path never steps."""


class PolynomialDecay(object):
    def __init__(self, *args, **kwargs):
        self.args = args
