mkdir -p gpurun_out/r02
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 2 --share-device --scaling strong --steps 40 --warmup 8 2>gpurun_out/r02/strong2.err | tail -n 1 > gpurun_out/r02/bench12_strong2_shared.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29562 bench.py --gpus 2 --share-device --steps 40 --warmup 8 2>gpurun_out/r02/weak2.err | tail -n 1 > gpurun_out/r02/bench12_weak2_shared.json
tail -n 3 gpurun_out/r02/strong2.err
python - <<'PY'
import json
for f in ('bench12_strong2_shared','bench12_weak2_shared'):
    try:
        d=json.load(open(f'gpurun_out/r02/{f}.json'))
        print(f, round(d['value']), round(d['ms_per_step'],4), d['scaling'], d['n_gpus'], d['config']['workload'][:90], '|', d['config']['exchange'][:100])
    except Exception as e: print(f, 'ERR', e, open(f'gpurun_out/r02/{f}.json').read()[:300])
PY
