export TMPDIR=/tmp
for v in new base new base; do
  L=$PWD/roreg_amd/libroreg_hip.so; [ $v = base ] && L=$PWD/roreg_amd/libroreg_hip_base.so
  rm -rf /tmp/kt_$v
  ROREG_HIP_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/kt_$v.json 2>/dev/null
  db=$(find /tmp/kt_$v -name '*.db' | head -1)
  python3 tools/rocprof_summary.py $db /tmp/kt_$v.txt > /dev/null
  echo "$v $(grep -E 'et_gather_batch' /tmp/kt_$v.txt | awk '{print $2, $3}') | ft_in $(grep -E 'ft_nonlin_kernel<true, false, 2, 4, 1, false>' /tmp/kt_$v.txt | awk '{print $2}' | head -1) | total $(head -1 /tmp/kt_$v.txt | awk '{print $(NF-1)}')"
done
