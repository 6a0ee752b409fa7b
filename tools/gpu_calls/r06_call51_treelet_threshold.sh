for i in 1 2; do timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 --cold-job-spp 16 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('cfg 4', d['value'], [l[:330] for l in d['cold_job']['trees'] if l.startswith('upload')])"; done
timeout 300 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --per-frame-frames 0 --surface-area-fold-steps 0 --cold-job-spp 16 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('cfg 5', d['value'], [l[:200] for l in d['cold_job']['trees'] if l.startswith('upload')])"
