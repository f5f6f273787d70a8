# usage (GPU box): bash tools/variant_suites.sh [tag] [soak] — the whole -m gpu suite under every structure / builder / storage variant the library can be
# switched to, and with the library built under -DPT_EXACT_RAY_RCP (IEEE divisions for the ray's reciprocal direction instead of v_rcp_f32: ADVICE r5;
# build it first on the authoring box: bash tools/build_variant.sh exact_rcp -DPT_EXACT_RAY_RCP); "soak": tests/soak_restarts.py afterwards
tag=${1:-r06}
for v in PTAMD_BVH4 PTAMD_RADIX_TREE PTAMD_BVH_LEGACY PTAMD_TWO_LEVEL PTAMD_NO_PAIRS PTAMD_TEX_NATIVE; do
  env $v=1 timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/suite_${tag}_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/suite_${tag}_$v.log)"
done
if [ -f platinum_amd/csrc/libptamd_exact_rcp.so ]; then
  env PTAMD_LIB=$GRAFT_REPO_ROOT/platinum_amd/csrc/libptamd_exact_rcp.so timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/suite_${tag}_exact_rcp.log 2>&1
  echo "PT_EXACT_RAY_RCP: $(tail -1 gpurun_out/suite_${tag}_exact_rcp.log)"
fi
if [ "${2:-}" = soak ]; then timeout -k 10 400 python tests/soak_restarts.py 180 > gpurun_out/soak_${tag}.log 2>&1; tail -2 gpurun_out/soak_${tag}.log | cut -c1-300; fi
