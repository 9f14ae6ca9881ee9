"""TEST INFRASTRUCTURE: CPU restatement of the reference's algorithms (maed_ref.py, loss_ref.py), the shims that let the reference itself
run in the build container (ref_shims.py) and the scripts that generated tests/golden/ from it (make_golden*.py).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; nothing under maed_amd/ does."""
