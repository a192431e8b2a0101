"""Per-kernel resource usage of the library as hipcc builds it (no GPU needed):
    python tools/resource_usage.py [extra -D flags ...]
VGPRs / AGPRs / SGPRs / scratch bytes per lane / occupancy / LDS per kernel, from -Rpass-analysis=kernel-resource-usage."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = [os.path.join(root, 'smplfitter_amd/csrc', f) for f in ('smplfit_hip.hip', 'sf_tables.cpp')]
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wno-unused-value',
       '-DSMPLFIT_BUILD_ID="res"', *src, '-o', '/tmp/_res_usage.so', '-Rpass-analysis=kernel-resource-usage', *sys.argv[1:]]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
keys = [('vgpr', r'VGPRs'), ('agpr', r'AGPRs'), ('sgpr', r'SGPRs'), ('scratch', r'ScratchSize \[bytes/lane\]'),
        ('occ', r'Occupancy \[waves/SIMD\]'), ('lds', r'LDS Size \[bytes/block\]')]
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split('\n')[0].strip()
    try:
        name = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'\(.*$', '', name)[:60]
    vals = []
    for label, pat in keys:
        m = re.search(pat + r': (\d+)', b)
        vals.append(f'{label} {m.group(1) if m else "?":>5}')
    print(f'{name:62s}', ' '.join(vals))
