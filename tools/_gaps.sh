cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace -d /tmp/kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/kt_line.json 2> /tmp/kt.err
db=$(find /tmp/kt -name '*.db' | head -1)
python3 -c "
import sqlite3,sys
db=sqlite3.connect('$db'); print([r[1] for r in db.execute('pragma table_info(kernels)')])"
python3 tools/rocprof_gaps.py $db 5.0
