"""Instruction histogram / register use of the kernels in a hipcc -save-temps assembly file (not a test)."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):', s, re.M):
    name = m.group(1)
    if pat not in name: continue
    start = m.end()
    end = s.find('.Lfunc_end', start)
    if end < 0: continue
    body = s[start:end]
    tail = s[end:end + 9000]
    g = lambda r: (re.search(r, tail).group(1) if re.search(r, tail) else None)
    ins = [l.split()[0] for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = Counter(ins)
    print(name, 'vgpr', g(r'; NumVgprs: (\d+)'), 'sgpr', g(r'; NumSgprs: (\d+)'), 'scratch', g(r'; ScratchSize: (\d+)'), 'occupancy', g(r'; Occupancy: (\d+)'), 'instrs', len(ins))
    print('   ', dict(c.most_common(45)))
