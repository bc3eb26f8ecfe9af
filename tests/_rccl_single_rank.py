"""Run by test_gpu_rccl.py in a subprocess with CAPI_RCCL_FORCE=1: every collective wrapper of the C-ABI goes through the
real librccl on a 1-rank communicator (the only RCCL configuration a 1-GPU box allows): binding, signatures, stream use."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi

assert os.environ.get("CAPI_RCCL_FORCE"), "run with CAPI_RCCL_FORCE=1"
L = capi.load()
h = capi.Handle(0)
torch_rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
if os.path.exists(torch_rccl):
    assert L.capi_comm_load_rccl(torch_rccl.encode()) == 0
uid = (C.c_char * 128)()
assert L.capi_comm_unique_id(C.cast(uid, C.c_void_p)) == 0, "ncclGetUniqueId"
comm = C.c_void_p()
rc = L.capi_comm_init_rank(C.byref(comm), h.h, 1, C.cast(uid, C.c_void_p), 0)
assert rc == 0, f"capi_comm_init_rank -> {rc}: {L.capi_last_error(h.h).decode()}"
n = 1 << 16
x = torch.arange(n, dtype=torch.float64, device="cuda") * 0.5 + 1.0
ref = x.clone()
y = torch.zeros(n, dtype=torch.float64, device="cuda")
stg = torch.zeros(n, dtype=torch.float64, device="cuda")
def ck(rc, what):
    assert rc == 0, f"{what} -> {rc}: {L.capi_last_error(h.h).decode()}"
ck(L.capi_bcast(comm, C.c_void_p(x.data_ptr()), n, 0), "bcast")
ck(L.capi_allreduce_sum(comm, C.c_void_p(x.data_ptr()), n), "allreduce")
ck(L.capi_reduce_sum(comm, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), n, 0), "reduce")
h.sync()
assert torch.equal(x, ref) and torch.equal(y, ref)
y.zero_()
ck(L.capi_allgather(comm, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), n), "allgather")
ck(L.capi_sendrecv_replace(comm, C.c_void_p(x.data_ptr()), n, 0, C.c_void_p(stg.data_ptr())), "sendrecv_replace")
h.sync()
assert torch.equal(y, ref) and torch.equal(x, ref) and torch.equal(stg, ref)
# the multi-path transfer wrapper (csrc/pair_paths.h over ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd): the one rank sends to itself
y.zero_()
dst = (C.c_int * 1)(0)
ck(L.capi_pairs_transfer(comm, dst, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), n, None), "pairs_transfer (self, through RCCL)")
h.sync()
assert torch.equal(y, ref)
dst[0] = -1
ck(L.capi_pairs_transfer(comm, dst, None, None, n, None), "pairs_transfer (nothing to send)")
child = C.c_void_p()
ck(L.capi_comm_split(comm, 0, 0, C.byref(child)), "comm_split")
r, s = C.c_int(-1), C.c_int(-1)
L.capi_comm_rank(child, C.byref(r)); L.capi_comm_size(child, C.byref(s))
assert (r.value, s.value) == (0, 1)
ck(L.capi_allreduce_sum(child, C.c_void_p(x.data_ptr()), n), "allreduce on the split communicator")
# the second stream carries collectives too (chunked SUMMA pipeline)
ck(L.capi_stream_select(h.h, 1), "stream_select")
ck(L.capi_bcast(child, C.c_void_p(x.data_ptr()), n, 0), "bcast on stream 1")
ck(L.capi_event_record(h.h, 5), "event_record")
ck(L.capi_stream_select(h.h, 0), "stream_select")
ck(L.capi_event_wait(h.h, 5), "event_wait")
h.sync()
assert torch.equal(x, ref)
ck(L.capi_comm_destroy(child), "destroy child")
ck(L.capi_comm_destroy(comm), "destroy")
print("rccl single-rank ok")
