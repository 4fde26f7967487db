"""TEST INFRASTRUCTURE — loads the reference's own Python files, unmodified, from
/root/reference with the stand-in modules of oracle/shim on sys.path (tensorflow, bezier, gym,
traci, sumolib are all absent from this image).  Container-only: the GPU box has
no /root/reference, and only oracle/gen_golden.py and the (auto-skipped) cross-check tests use this.
"""
import os
import sys

REF = '/root/reference'
SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'shim')


def available():
    return os.path.isfile(os.path.join(REF, 'dynamics_and_models.py'))


def load():
    if not available():
        raise RuntimeError('reference tree not present (expected only inside the build container)')
    os.environ.setdefault('SUMO_HOME', '/nonexistent-sumo')
    os.environ.setdefault('MPLBACKEND', 'Agg')
    for p in (SHIM, REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    import dynamics_and_models as DAM
    import endtoend_env_utils as UTL
    import endtoend as E2E
    import traffic as TRF
    return DAM, UTL, E2E, TRF
