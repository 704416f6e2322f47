"""CPU oracle for the GPS-Gaussian render hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product path (gps-gaussian_amd/) never does: it fails loudly when the HIP library is missing.
"""
