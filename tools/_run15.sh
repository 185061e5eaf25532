for v in 0 1; do echo "== HIP_FORCE_DEV_KERNARG=$v"; HIP_FORCE_DEV_KERNARG=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-context 2>/dev/null | python -c "
import sys, json
p = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(p['value'], p['ms_per_step'], p['ms_per_step_device']['median'], p['stages_ms_per_step'], p['parity']['max_abs'])
"; done
echo "== unset"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-context 2>/dev/null | python -c "
import sys, json
p = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(p['value'], p['ms_per_step'], p['ms_per_step_device']['median'], p['stages_ms_per_step'])
"
strings /opt/rocm/lib/libamdhip64.so | grep -i "KERNARG" | head
