#!/bin/bash
# A/B of the Fiat-Shamir channel: host (CSTARK_HOST_CHANNEL=1) against device, un-profiled bench lines and the idle-time report of a kernel trace
set -o pipefail
O=gpurun_out
for rep in 1 2; do
  CSTARK_HOST_CHANNEL=1 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs > $O/chan_host_$rep.json 2>> $O/chan_ab.err || exit 1
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs > $O/chan_dev_$rep.json 2>> $O/chan_ab.err || exit 1
done
python3 - <<'PY'
import json
for k in ("host", "dev"):
    for rep in (1, 2):
        d = json.loads(open("gpurun_out/chan_%s_%d.json" % (k, rep)).read().strip().splitlines()[-1])
        print(k, rep, "ms_per_step", d["ms_per_step"], "two_in_flight", d.get("two_in_flight", {}).get("proofs_per_s") if isinstance(d.get("two_in_flight"), dict) else d.get("two_in_flight"))
PY
bash tools/gpu_jobs/timeline.sh r04_host_channel CSTARK_HOST_CHANNEL=1 && bash tools/gpu_jobs/timeline.sh r04_dev_channel
