"""Empty stand-in for ASE (absent in the build container); only what the reference's
descriptor/kernel/regression modules touch at import time. Used ONLY by make_golden.py."""
