timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "cluster or jorg or houdayer or sw or wolff or cmr or fk" > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
export PEAPODS_DEVELOPER=1 PEAPODS_STREAMS=1
for m in jorg houdayer; do
  PEAPODS_SW_STAMPS=$PWD/gpurun_out/sw_$m.bin timeout -k 10 200 python bench.py --config c4 --oc-mode $m --oc-variant sw --steps 150 --warmup 20 --no-cpu-baseline --repeats 1 > gpurun_out/b_s$m.json 2>gpurun_out/b_s$m.err
  echo $m; python tools/_sw_stamps.py gpurun_out/sw_$m.bin
  python -c "
import json
r=json.loads(open('gpurun_out/b_s$m.json').read().strip().splitlines()[-1]); print(r['ms_per_step'], [(n,round(v['ms_per_step']*1000,1)) for n,v in r['roofline']['kernels_per_step'].items() if 'pair_stream' in n or 'cluster' in n])"
done
