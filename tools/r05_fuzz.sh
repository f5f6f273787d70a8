# usage (GPU box): bash tools/r05_fuzz.sh <first> <chunks> [extras] — chunks of 500 seeded random scenes through the HIP path against the oracle (tests/fuzz_parity_sweep.py)
first=$1; chunks=$2; extras=${3:-}
for c in $(seq 0 $((chunks - 1))); do
  timeout -k 10 900 python tests/fuzz_parity_sweep.py $((first + 500 * c)) 500 $extras 2>&1 | tail -1 | tee -a gpurun_out/fuzz_${first}_${extras:-plain}.jsonl | cut -c1-400
done
