#!/bin/bash
D=gpurun_out/$1
tail -2 $D/pytest.log 2>/dev/null
python - <<PY
import json
for l in open("$D/bench.json"):
    if l.startswith("{"):
        d=json.loads(l); print("BENCH %.1f M pk/s  %.2f us/launch  frac %.3f  %s  %s" % (d['value']/1e6, d['roofline']['launch_ms']*1e3, d['roofline']['frac'], d['config']['kernels'], d['config']['parity']))
PY
grep -v "Exception ignored\|Traceback\|File \"\|AttributeError\|TypeError\|amdgpu.ids\|stamps of wave" $D/stamps.txt | tail -12 | cut -c1-175
