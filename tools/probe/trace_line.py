import json,sys
for ln in sys.stdin:
    ln=ln.strip()
    if not ln.startswith('{'): continue
    d=json.loads(ln)
    print('value',d['value'],d['block_values'],'gap',d['collect_gap_ms']['max'],d['collect_gap_ms']['argmax'])
    t=d.get('pipeline_trace') or {}
    for k,v in t.items():
        if isinstance(v,dict) and 'max' in v: print('  %-28s p50 %.4f max %.4f at pair %s'%(k,v['p50'],v['max'],v.get('argmax_pair')))
