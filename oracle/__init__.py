"""TEST INFRASTRUCTURE (not product code): CPU oracles for the slam_toolbox hot path.

oracle.karto   -- ctypes binding of the plain-C restatement (karto_oracle.c, occupancy_oracle.c); oracle.spa is the numpy / scipy restatement of the solver
oracle.ref     -- ctypes binding of the reference's own sources built in place (oracle/_ref)
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
