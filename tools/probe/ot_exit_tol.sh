# iterations the bench's pairs run under multiples of the early-exit tolerance (ROREG_OT_EXIT_TOL)
for t in 1 2 4 8 16 32; do
  echo "tol x $t"; ROREG_OT_EXIT_TOL=$t python3 tools/probe/sinkhorn_convergence.py 5000 6 2>&1 | grep "iteration stats"
done
