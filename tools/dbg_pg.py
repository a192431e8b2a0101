"""Isolation experiment: k_pair_gram_bm (time hook, id 6) repeatedly on stream A while the posedirs GEMM
(id 2) runs in a loop on stream B from another host thread.  Is gramP stable?"""
import ctypes as C, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dbg_ws.py')).read().split("nbytes = h.workspace_bytes(B)")[0])
from smplfitter_amd import _lib
lib = _lib.load()
Bc = 2048
os.environ['SMPLFIT_CHUNKS'] = '1'
nbytes = h.workspace_bytes(Bc)
wsA = torch.zeros(nbytes, dtype=torch.uint8, device=dev); wsB = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
f.fit(tv[:Bc], tj[:Bc], num_iter=2, beta_regularizer=1.0, _workspace=wsA)
f.fit(tv[Bc:], tj[Bc:], num_iter=2, beta_regularizer=1.0, _workspace=wsB)
torch.cuda.synchronize()
regs, per = regions(Bc)
off = {n: (o, s) for n, o, s in regs}
sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
stop = False
def neighbour(kid):
    ms = C.c_float()
    while not stop:
        lib.smplfit_time_kernel_f32(h.ptr, kid, Bc, 20, C.c_void_p(wsB.data_ptr()), nbytes, C.c_void_p(sB.cuda_stream), C.byref(ms))
def probe(kid, region, n=30):
    ms = C.c_float(); snaps = []
    o, s = off[region]
    for i in range(n):
        wsA[o:o+s] = 0
        lib.smplfit_time_kernel_f32(h.ptr, kid, Bc, 1, C.c_void_p(wsA.data_ptr()), nbytes, C.c_void_p(sA.cuda_stream), C.byref(ms))
        torch.cuda.synchronize()
        snaps.append(wsA[o:o+s].clone())
    nd = [int((x != snaps[0]).sum()) for x in snaps[1:]]
    return nd
os.environ['SMPLFIT_GEMM'] = 'bf16x3'
for nb_kid, nb_name in ((None, 'alone'), (2, 'gemm')):
    stop = False
    th = None
    if nb_kid is not None:
        th = threading.Thread(target=neighbour, args=(nb_kid,)); th.start()
    r1 = probe(6, 'gramP', 60); r3 = probe(5, 'psumP', 60)
    stop = True
    if th: th.join()
    print(f'lib={os.environ.get("SMPLFIT_LIB","in-tree")[-22:]} neighbour={nb_name:6s} pair_gram bytes differing: {sum(r1)} (max {max(r1)}); lbs: {sum(r3)}', flush=True)
