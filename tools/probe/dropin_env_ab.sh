# usage (on the GPU box): bash tools/probe/dropin_env_ab.sh   -- the evaluator's engine route under runtime copy-path settings (one box, back to back)
mkdir -p gpurun_out/r06
for setting in "NONE=1" "DEBUG_CLR_LIMIT_BLIT_WG=16" "DEBUG_CLR_LIMIT_BLIT_WG=64" "GPU_BLIT_ENGINE_TYPE=2" "GPU_FORCE_BLIT_COPY_SIZE=0" "NONE=2"; do
  echo "== $setting"
  env $setting timeout 300 python3 tools/probe/dropin_timeline.py 2>&1 | grep "pairs/s" | cut -c1-60
done 2>&1 | tee gpurun_out/r06/dropin_env_ab.txt
