# span Spectrogram with 8-column tiles on 512 threads (two workgroups per CU) against the 16-column kernel, same box, alternating
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
L=cyberether_amd/lib/libjetstream_hip.so
cp $L /tmp/base.so
for rep in 1 2; do for v in base span_tw8_c2 span_tw8_c4; do
  if [ $v = base ]; then cp /tmp/base.so $L; else cp cyberether_amd/lib/variants/$v.so $L; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-configs --no-host-fed 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['ms_per_step']*1e3,3), 'us/step', {k:round(v*1e3,2) for k,v in d['config']['units_ms'].items()}, 'parity', d['parity'].get('bit_exact'), d['parity'].get('spectrogram_state_bit_exact'))"
done; done
cp /tmp/base.so $L
