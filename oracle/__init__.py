"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the GL-Fusion hot path.

Nothing in the product package (gl-fusion_amd/) may import this package.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only
as the checker.  See oracle/glfusion_ref.py for the pinning status.
"""
