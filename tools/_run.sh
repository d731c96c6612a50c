timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
  timeout 300 python bench.py 2>/dev/null | tail -1 > /tmp/b.json
  python3 -c "
import json
d=json.load(open('/tmp/b.json')); h=d['host_scope_ms_per_frame']
print(d['value'], d['ms_per_step'], d['ate_rmse_m'], h['localize'], h['refine_subwindow'], h['refine_window'], h['slide_window'])"
done
