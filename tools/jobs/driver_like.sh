#!/bin/bash
# What the driver runs for SCALE_rNN, on ONE GPU: every rank claims a host of its own (SLGC_RANKS_AS_HOSTS=1: RCCL over the loopback socket
# transport), launched by torch.distributed.run exactly like the driver does.  A correctness run of the complete N > 1 path (main strategy,
# compute-only leg, self-verification, throughput mode, alternative exchanges), never a measurement.
O=gpurun_out/driver_like; mkdir -p $O
export SLGC_RANKS_AS_HOSTS=1 SLGC_BENCH_TIMEOUT_S=400 SLGC_BENCH_ALT_TIMEOUT_S=200
for n in 2 4 8; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 20 --warmup 5 > $O/n$n.out 2> $O/n$n.err; rc=$?
  python3 - $O/n$n.out $rc $n <<'PY'
import json, sys
lines = [ln for ln in open(sys.argv[1]) if ln.startswith("{")]
if not lines:
    print("N", sys.argv[3], "rc", sys.argv[2], "NO JSON"); sys.exit(0)
j = json.loads(lines[-1])
# (round 5: the printed line is the compact object -- verify_ok, sharded.alternatives, throughput_mode_value)
print("N", sys.argv[3], "rc", sys.argv[2], "json lines", len(lines), "bytes", len(lines[-1]), "| value", j.get("value"), "| scaling", j.get("scaling"), "| verify_ok", j.get("verify_ok"),
      "| sharded", j.get("sharded"), "| thr", j.get("throughput_mode_value"), "| err", j.get("error"))
PY
  tail -2 $O/n$n.err | cut -c1-300
done
