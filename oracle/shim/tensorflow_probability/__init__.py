"""TEST INFRASTRUCTURE — utils/policy.py:20-22 binds tfp.distributions / tfp.bijectors in its class body; only the
stochastic branch (deterministic_policy = False) would use them, and the fixtures do not take it."""
import types

distributions = types.SimpleNamespace()
bijectors = types.SimpleNamespace()
