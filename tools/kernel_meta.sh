#!/bin/bash
# per-kernel VGPR / scratch metadata of every gfx950 code object in a shared library: tools/kernel_meta.sh lib.so [regex]
lib=$(readlink -f $1); filt=${2:-.}
tmp=$(mktemp -d); cd $tmp
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=fat.bin $lib
python3 - <<'PY'
import re
data=open('fat.bin','rb').read()
idx=[m.start() for m in re.finditer(b'__CLANG_OFFLOAD_BUNDLE__',data)]
for n,i in enumerate(idx):
    open('b%d.bin'%n,'wb').write(data[i:(idx[n+1] if n+1<len(idx) else len(data))])
PY
for b in b*.bin; do
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$b --output=$b.co --unbundle 2>/dev/null || continue
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $b.co
done | python3 -c "
import sys,re
cur={}
out=set()
for line in sys.stdin:
    m=re.match(r'\s*(?:- )?\.(agpr_count|name|private_segment_fixed_size|vgpr_count|vgpr_spill_count):\s*(\S+)',line)
    if not m: continue
    if m.group(1)=='agpr_count' and cur.get('name'):
        cur={}
    cur[m.group(1)]=m.group(2)
    if len(cur)==5:
        n=cur['name'].replace('_ZN2rt7k_stageIN3bbs','')[:60]
        out.add('%-62s vgpr=%s agpr=%s spill=%s scratch=%s'%(n,cur['vgpr_count'],cur['agpr_count'],cur['vgpr_spill_count'],cur['private_segment_fixed_size']))
        cur={}
for l in sorted(out): print(l)
" | grep -E "$filt"
rm -rf $tmp
