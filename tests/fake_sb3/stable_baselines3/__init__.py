"""TEST DOUBLE -- not stable-baselines3.  A minimal package with the import path and the abstract ``VecEnv`` class
surface of SB3 1.6 (the reference's pinned version, requirements.txt:82), written for tests/test_sb3_contract.py so
that "BatchedRocket6DOF derives from SB3's VecEnv when SB3 is importable" can be checked in an image without SB3."""
__version__ = "1.6.0-testdouble"
