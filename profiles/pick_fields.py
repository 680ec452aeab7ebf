import sys,json
for l in sys.stdin:
    l=l.strip()
    if not l.startswith("{"): continue
    d=json.loads(l)
    print({k:d[k] for k in ("keys","table_bytes","kernel_ms","Gbases_per_s","reads","sample_differ","sample_equal","spilled_keys","args","genome_bases","layout","same_answers","gbases_per_s","check") if k in d})
