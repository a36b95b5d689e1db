import sys, re, subprocess
txt = sys.stdin.read()
cur = None
rows = []
for line in txt.splitlines():
    m = re.search(r'remark: (.*?) \[-Rpass', line)
    if not m: continue
    body = m.group(1)
    if body.startswith('Function Name:'):
        name = body.split(': ',1)[1]
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        cur = {'name': dem[:70]}
        rows.append(cur)
    elif cur is not None and ':' in body:
        k, v = body.split(':', 1)
        cur[k.strip()] = v.strip()
for r in rows:
    print(r['name'], '| VGPR', r.get('VGPRs'), 'AGPR', r.get('AGPRs'), 'SGPR', r.get('TotalSGPRs'), 'scratch', r.get('ScratchSize [bytes/lane]'), 'spillV', r.get('VGPR Spill'), 'occ', r.get('Occupancy [waves/SIMD]'), 'LDS', r.get('LDS Size [bytes/block]'))
