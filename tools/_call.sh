set -x
mkdir -p gpurun_out
for m in raw-lib-first raw-torch-first abi-lib-first; do timeout -k 10 300 python tools/runtime_order_probe.py $m > gpurun_out/rt_$m.log 2>&1; echo "exit $?" >> gpurun_out/rt_$m.log; done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t1_tests.log 2>&1; tail -3 gpurun_out/t1_tests.log
timeout -k 10 300 python bench.py --workload c3 --no-cpu-baseline --steps 8 > gpurun_out/t1_c3.json 2> gpurun_out/t1_c3.err && timeout -k 10 300 python bench.py --workload c2 --no-cpu-baseline --steps 8 > gpurun_out/t1_c2.json 2> gpurun_out/t1_c2.err
tail -2 gpurun_out/t1_c3.err
