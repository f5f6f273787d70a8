# usage (GPU box): bash tools/r05_variant_suites.sh — the whole -m gpu suite under every structure / builder / storage variant the library can be switched to, then the restart soak
for v in PTAMD_BVH4 PTAMD_RADIX_TREE PTAMD_BVH_LEGACY PTAMD_TWO_LEVEL PTAMD_NO_PAIRS PTAMD_TEX_NATIVE; do
  env $v=1 timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/suite_r05_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/suite_r05_$v.log)"
done
timeout -k 10 400 python tests/soak_restarts.py 180 > gpurun_out/soak_r05.log 2>&1; tail -2 gpurun_out/soak_r05.log | cut -c1-300
