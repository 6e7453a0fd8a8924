for cfg in "8 4" "8 6" "8 5" "4 8" "6 6" "16 3"; do set -- $cfg
  python bench.py --steps 20 --warmup 5 --scripted-steps 10 --on-policy-steps 10 --surface-steps 0 --latency-reps 0 --window-reps 0 --no-cpu-baseline --batched-envs $1 --batched-groups $2 > gpurun_out/grp_$1_$2.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/grp_$1_$2.json')); print('envs $1 groups $2: batched', d['batched']['value'], 'groups', d['batched_groups']['value'] if 'value' in d.get('batched_groups',{}) else d.get('batched_groups'))"
done
