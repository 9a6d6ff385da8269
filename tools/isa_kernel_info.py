import re, sys, subprocess
s=open(sys.argv[1]).read()
names=[]; rows=[]
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name=m.group(1); body=m.group(2)
    def g(k):
        r=re.search(r'\.amdhsa_'+k+r'\s+(\S+)', body); return r.group(1) if r else None
    names.append(name); rows.append((g('next_free_vgpr'),g('accum_offset'),g('next_free_sgpr'),g('private_segment_fixed_size')))
dn=subprocess.run(['c++filt']+names,capture_output=True,text=True).stdout.strip().split('\n')
for d,r in zip(dn,rows):
    d=d.replace('void tsp::','').split('(')[0]
    print(f"{d:60s} vgpr {r[0]:>4} accum_off {r[1]:>4} sgpr {r[2]:>4} scratch {r[3]:>5}")
