"""TEST INFRASTRUCTURE (not product code): CPU oracles for the slam_toolbox hot path.

oracle.karto   -- ctypes binding of the plain-C restatement (karto_oracle.c, spa_oracle.c)
oracle.ref     -- ctypes binding of the reference's own sources built in place (oracle/_ref)
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
