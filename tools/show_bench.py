import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            print(f)
            print(' value %.1f %s  ms/step %.1f  s/query %.2f decode %.2f ms/tok frames/s %.1f' % (d['value'], d['unit'], d['ms_per_step'], d['sec_per_query'], d['decode_ms_per_token'], d['frames_per_s']))
            print(' stages', {k: round(v, 1) for k, v in d['stage_ms_per_step'].items()})
            for k, v in d['kernel_families'].items():
                print('   ', k, {a: (round(b, 2) if isinstance(b, float) else b) for a, b in v.items()})
            if d.get('roofline'):
                print(' roofline', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d['roofline'].items() if k != 'kernel'})
            print(' first_token', d.get('first_token'), 'attn_gain', d.get('attn_gain'))
            if d.get('verify'):
                print(' verify', json.dumps(d['verify']))
            if 'cpu_baseline' in d:
                print(' cpu', d['cpu_baseline'].get('value'), d['cpu_baseline'].get('cores'), d.get('speedup_vs_cpu'))
