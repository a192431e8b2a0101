"""Neighbour-interaction probe (DESIGN.md section 4 "EXCLUSIVE CU"): a read-only LDS victim beside the library's own
split-bf16 GEMM in a debug build that lets other kernels onto the GEMM's CUs.

    tools/build_variant.sh gemm4 -DSMPLFIT_GEMM_WAVES=4 -DSMPLFIT_GEMM_SHARED_CU=1         (here)
    SMPLFIT_LIB=build_ab/libgemm4.so python tools/lds_probe.py                             (on the GPU box)

The victim (k_lds_victim) fills 8 KB of LDS once and then only reads it (uniform / per-lane addresses, 4 / 16 bytes),
checking every value; every wrong value is logged with lane, address, the value a second read returns, and the table
is verified at the end.  Run alone (control) and while another host thread loops the posedirs GEMM on a second stream."""
import ctypes as C, os, sys, threading, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import _lib, synth
from smplfitter_amd.pt import BodyFitter, BodyModel

kind = sys.argv[1] if len(sys.argv) > 1 else 'smpl'
dev = torch.device('cuda:0')
root = synth.ensure_model_root(kinds=(kind,))
m = BodyModel(kind, 'neutral', model_root=f'{root}/{kind}', num_betas=10, device=dev)
f = BodyFitter(m)
J = m.num_joints
B = 2048
rs = np.random.RandomState(42)
t = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev)
fw = m(t(rs.randn(B, 3 * J) * 0.1), t(rs.randn(B, 10) * 0.5), t(rs.randn(B, 3)))
tv, tj = fw['vertices'].contiguous(), fw['joints'].contiguous()
h = m._native(dev)
os.environ['SMPLFIT_CHUNKS'] = '1'
_lib.reload_options()
ws = torch.zeros(h.workspace_bytes(B), dtype=torch.uint8, device=dev)
f.fit(tv, tj, num_iter=2, beta_regularizer=1.0, _workspace=ws)
torch.cuda.synchronize()
lib = _lib.load()
lib.smplfit_debug_lds_victim.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
MAXLOG = 4096
log = torch.zeros(MAXLOG * 8, dtype=torch.int32, device=dev)
nlog = torch.zeros(1, dtype=torch.int32, device=dev)
stop = False


def neighbour(kid):
    ms = C.c_float()
    while not stop:
        lib.smplfit_time_kernel_f32(h.ptr, kid, B, 20, C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(sB.cuda_stream), C.byref(ms))


lib.smplfit_debug_lds_victim_fma.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
NW = 1024  # four one-wave workgroups per CU: the GEMM's workgroups fit beside them
outv = torch.zeros(NW * 32 * 64, dtype=torch.float32, device=dev)


def fma_victim(rewrite, reps):
    res = []
    for _ in range(reps):
        outv.zero_()
        assert lib.smplfit_debug_lds_victim_fma(C.c_void_p(sA.cuda_stream), NW, 60000, rewrite, C.c_void_p(outv.data_ptr())) == 0
        sA.synchronize()
        res.append(outv.clone())
    return res


KIND = {0: 'uniform b32', 1: 'per-lane b32', 2: 'uniform b128', 3: 'per-lane b128', 4: 'final contents'}
for name, kid in (('alone', None), ('beside the GEMM', 2), ('beside the LBS pass', 5)):
    stop = False
    th = None
    if kid is not None:
        th = threading.Thread(target=neighbour, args=(kid,))
        th.start()
    for rewrite in (0, 1):
        ref = fma_victim(rewrite, 1)[0] if kid is None else refs[rewrite]
        if kid is None:
            refs = globals().setdefault('refs', {})
            refs[rewrite] = ref
        runs = fma_victim(rewrite, 4)
        bad = [(r != ref) for r in runs]
        nb = sum(int(b.sum()) for b in bad)
        lanes = collections.Counter()
        for b in bad:
            idx = torch.nonzero(b).flatten().cpu().numpy()
            lanes.update((idx % 64).tolist())
        print(f'fma victim (rewrite {rewrite}) {name}: {nb} of {4 * ref.numel()} values differ from the solo run' + (f'; lanes {dict(sorted(lanes.items()))}' if nb else ''), flush=True)
    log.zero_(); nlog.zero_()
    torch.cuda.synchronize()
    for rep in range(3):
        assert lib.smplfit_debug_lds_victim(C.c_void_p(sA.cuda_stream), 1024, 400000, C.c_void_p(log.data_ptr()), C.c_void_p(nlog.data_ptr()), MAXLOG) == 0
    sA.synchronize()
    stop = True
    if th:
        th.join()
    torch.cuda.synchronize()
    n = int(nlog.item())
    rec = log.cpu().numpy().view(np.uint32).reshape(-1, 8)[:min(n, MAXLOG)]
    print(f'victim {name}: {n} wrong values in 3 x 1024 waves x 400000 iterations')
    if n:
        kinds = collections.Counter(int(r[3]) for r in rec)
        print('  by read:', {KIND[k]: v for k, v in kinds.items()})
        lanes = collections.Counter(int(r[2]) for r in rec)
        print('  by lane:', dict(sorted(lanes.items())))
        same = sum(1 for r in rec if r[7] == r[6])
        print(f'  second read of the same cell correct again: {same} of {len(rec)}; still wrong: {len(rec) - same}')
        for r in rec[:12]:
            print(f'    wg {r[0]} it {int(np.int32(r[1]))} lane {r[2]} {KIND[int(r[3])]} addr {r[4]} got {r[5]:#x} expected {r[6]:#x} again {r[7]:#x}')
