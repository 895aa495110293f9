import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print(d['label'], 'span %.1f slowest %.1f mean %.1f' % (d['mean_span_us'], d['mean_slowest_us'], d['mean_instance_us']), 'fit', {k: round(v, 2) for k, v in d['fit_us'].items()},
              '\n   slowest', {k: round(v, 1) for k, v in d['phases_us_slowest'].items()}, '\n   mean', {k: round(v, 1) for k, v in d['phases_us_mean'].items()},
              '\n   setup/tail', {k: round(v, 2) for k, v in d.get('setup_and_tail_us_mean', {}).items()}, 'runfit', d['runs_fit_us(const, per_regular_op)'], 'warmfit', d['warm_start_fit_us(const, per_op)'], 'noop %.2f' % d['run_without_operation_us'])
    elif l.startswith('HDSM_PROFILE'):
        print(l.strip()[:1500])
    elif 'gpurun]' in l and ('status' in l or 'GPU-min' in l):
        print(l.strip())
    elif l.startswith('HDSM_'):
        print(l.strip())
