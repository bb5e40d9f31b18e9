set -e
export RT_BENCH_REHEARSE_COLLECTIVES=1
run() { echo "== $1"; shift; env "$@" timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/reh.json 2> gpurun_out/reh.err || echo "rc $?"; python3 -c "
import json
d=json.load(open('gpurun_out/reh.json'))
print(d['ms_per_step'], d['config']['multi_gpu_path'], d['config'].get('library_path_error'), d.get('multi_gpu_frame_equal'))"; }
run plain A=1
run hang RT_BENCH_TEST_REHEARSAL=hang RT_BENCH_REHEARSAL_SECONDS=15
run error RT_BENCH_TEST_REHEARSAL=error
run differ RT_BENCH_TEST_REHEARSAL=differ

