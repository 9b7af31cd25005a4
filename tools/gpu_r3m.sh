#!/bin/bash
python bench.py --strong --steps 5 --warmup 2 --single-datapath --no-cpu-baseline --no-eager-baseline --no-gate 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['config']['global_batch_rays'], d['scaling'])"
