cd $GRAFT_REPO_ROOT
for s in 16 32 64; do
python bench.py --slots $s --no-cpu-baseline --no-host-fed --no-configs --no-alt 2>/dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('slots', $s, 'step_us', round(b['ms_per_step']*1e3,3), 'value', round(b['value']), 'kernel_ms', round(b['roofline']['kernel_ms']*1e3,2), 'cycles/launch', b['roofline']['cycles_per_launch'], 'frac', round(b['roofline']['frac'],4), 'step_frac', round(b['roofline']['step_frac'],4), 'parity', b['parity']['bit_exact'])"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp64
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp64 -- python $GRAFT_REPO_ROOT/bench.py --slots 64 --no-cpu-baseline --no-host-fed --no-configs --no-alt --no-parity --min-time 0.05 > /tmp/rp64.log 2>&1
python $GRAFT_REPO_ROOT/tools/kstats.py /tmp/rp64 | head -4
