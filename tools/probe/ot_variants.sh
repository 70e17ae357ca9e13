for v in 0 1 2 3 4; do echo "== variant $v"; ROREG_OT_VARIANT=$v python tools/time_sinkhorn.py 2500 30 2>&1 | grep P=; done
