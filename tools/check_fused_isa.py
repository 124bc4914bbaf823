#!/usr/bin/env python3
"""Audit of the single-launch transform's ISA (ntt_fused.cuh) -- run after every change to that file:
  * no scratch, registers allocated (VGPR + AGPR, rounded to 8) fit two waves per SIMD,
  * the compiler itself uses no accumulator register (a0..a2 carry the asynchronous control results),
  * prints the skeleton (memory operations, waits, barriers, branches) of the kernel for reading.
usage: python3 tools/check_fused_isa.py [--skeleton]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.gettempdir(), "pha_fused.s")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "--offload-device-only", "-S", "-o", out,
       *os.environ.get("PHA_EXTRA_DEFINES", "").split(), os.path.join(ROOT, "phantom-fhe_amd", "csrc", "phantom_amd.hip")]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
ok = True
for m in re.finditer(r'\.amdhsa_kernel (\S*f(?:ntt|intt)_fused\S*)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, desc = m.group(1), m.group(2)
    g = lambda k: int(re.search(k + r'\s+(\S+)', desc).group(1))
    total, scratch = g(r'\.amdhsa_next_free_vgpr'), g(r'\.amdhsa_private_segment_fixed_size')
    alloc = (total + 7) // 8 * 8
    i = s.index(name + ":")
    body = s[i:s.index('s_endpgm', i)].split('\n')
    inasm, bad = False, 0
    for l in body:
        if '#ASMSTART' in l: inasm = True
        elif '#ASMEND' in l: inasm = False
        elif not inasm and re.search(r'accvgpr|\ba\[?\d', l.split(';')[0]): bad += 1
    good = scratch == 0 and alloc * 2 <= 512 and bad == 0
    ok &= good
    print(f"{'ok ' if good else 'BAD'} {name[:70]}: registers {total} (allocated {alloc}, {512 // alloc} waves/SIMD), "
          f"accum_offset {g(r'.amdhsa_accum_offset')}, scratch {scratch}, compiler AGPR uses {bad}, {len(body)} lines")
    if '--skeleton' in sys.argv:
        keep = re.compile(r'^\s*(global_|buffer_|flat_|s_barrier|s_waitcnt|s_cbranch|s_branch|\.LBB|s_load|s_sleep|;;#ASM|scratch_|v_accvgpr)')
        for k, l in enumerate(body):
            if keep.match(l): print(f'{k:6d} {l.strip()[:120]}')
        break
sys.exit(0 if ok else 1)
