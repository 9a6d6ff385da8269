"""CPU oracle for the SPH splat + colormap hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (topsy_amd) never does.
"""
