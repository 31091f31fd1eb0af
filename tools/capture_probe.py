#!/usr/bin/env python3
"""Which captured node kills hipGraph instantiation?  (round-1 abort: `Fatal Python error: Segmentation fault` in
torch/cuda/graphs.py capture_end, one-rank sharded rehearsal, gpurun_out/fd2/psdp2.log.)

Each case runs in its OWN child process (a crash costs the child), captures with keep_graph=True so that capture_end
does NOT instantiate, lists the graph's node types through hipGraphGetNodes / hipGraphNodeGetType (ctypes on
libamdhip64), then instantiates and replays.  Cases: a contiguous device-to-device copy_ (a memcpy node) of several
sizes on the capture stream and on a side stream; a strided copy (a kernel node); the one-rank RCCL all-to-all (RCCL
substitutes a memcpy for a self-addressed peer); two memcpy nodes of 25.7 MB in one graph (the round-1 flow).

    python tools/capture_probe.py            # runs every case, prints one line each
    python tools/capture_probe.py CASE       # (child)
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NODE_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "waitEvent", 7: "eventRecord",
              8: "extSemSignal", 9: "extSemWait", 10: "memAlloc", 11: "memFree", 12: "memcpyFromSymbol", 13: "memcpyToSymbol"}
CASES = ["memcpy_1MB", "memcpy_8MB", "memcpy_26MB", "memcpy_64MB", "memcpy_26MB_side", "memcpy_2x26MB_side", "strided_26MB",
         "rccl_self_alltoall_26MB", "rccl_self_alltoall_26MB_side", "rccl_allreduce_inplace_1MB", "rccl_reducescatter_inplace_1MB",
         "rccl_allgather_1MB", "rccl_group_1MB", "rccl_two_comms_1MB", "rccl_stageflow_26MB_side"]


def node_types(g):
    import torch
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    raw = ctypes.c_void_p(g.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    if hip.hipGraphGetNodes(raw, None, ctypes.byref(n)) != 0:
        return ["?"]
    arr = (ctypes.c_void_p * n.value)()
    hip.hipGraphGetNodes(raw, arr, ctypes.byref(n))
    out = []
    for i in range(n.value):
        t = ctypes.c_int(-1)
        hip.hipGraphNodeGetType(ctypes.c_void_p(arr[i]), ctypes.byref(t))
        out.append(NODE_TYPES.get(t.value, str(t.value)))
    return out


def child(case):
    import torch
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream(dev)
    on_side = case.endswith("_side")
    mb = {"1MB": 1, "8MB": 8, "26MB": 25.7, "64MB": 64}[[p for p in case.split("_") if p.endswith("MB")][0].replace("2x", "")]
    n = int(mb * 2**20 / 4) // 1024 * 1024
    a, b, c = (torch.randn(n, device=dev) for _ in range(3))
    comm = None
    if case.startswith("rccl"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
        from r3d_amd.rccl import RcclComm
        comm = RcclComm()
        comm2 = RcclComm() if ("two_comms" in case or "stageflow" in case) else None

    def body():
        if case.startswith("memcpy_2x"):
            b.copy_(a)
            c.copy_(b)
        elif case.startswith("memcpy"):
            b.copy_(a)
        elif case.startswith("strided"):
            b.view(1024, -1).copy_(a.view(-1, 1024).t())
        elif case.startswith("rccl_self_alltoall"):
            comm.all_to_all(b.view(1, -1), a.view(1, -1))
        elif case.startswith("rccl_allreduce"):
            comm.all_reduce(a)
        elif case.startswith("rccl_reducescatter"):
            comm.reduce_scatter_inplace(a.view(-1, 128))
        elif case.startswith("rccl_allgather"):
            comm.all_gather(b.view(-1, 128), a.view(-1, 128))
        elif case.startswith("rccl_group"):
            with comm.group():
                comm.all_gather(b.view(-1, 128), a.view(-1, 128))
                comm.all_reduce(c)
        elif case.startswith("rccl_two_comms"):
            comm.all_reduce(a)
            comm2.all_reduce(c)
        else:       # the round-1 side graph: denominator (tiny kernel + all-reduce), contiguous copy_, self all-to-all
            den = c[:1]
            torch.mul((a[:64] != 3.0).sum().to(torch.float32).reshape(1), 1.0, out=den)
            comm2.all_reduce(den)
            b.view(1, -1).copy_(a.view(-1, 1).transpose(0, 1))
            comm2.all_to_all(c.view(1, -1), b.view(1, -1))
    body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    if on_side:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(g, stream=side):
            body()
    else:
        with torch.cuda.graph(g):
            body()
    types = node_types(g)
    print(f"[{case}] captured nodes: {types}", flush=True)
    g.instantiate()
    print(f"[{case}] instantiated", flush=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ok = bool(torch.equal(b, a)) if case.startswith(("memcpy", "rccl_self", "rccl_allgather", "rccl_group", "rccl_stageflow")) else True
    print(f"[{case}] replayed x3, result {'ok' if ok else 'WRONG'}", flush=True)
    os._exit(0)


def main():
    if len(sys.argv) > 1:
        return child(sys.argv[1])
    for case in CASES:
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), case], capture_output=True, text=True, timeout=120)
            lines = [l for l in (r.stdout + r.stderr).splitlines() if l.startswith(f"[{case}]") or "Fatal" in l or "Error" in l]
            print(f"{case}: rc={r.returncode} | " + " | ".join(lines[-4:]), flush=True)
        except subprocess.TimeoutExpired:
            print(f"{case}: TIMEOUT", flush=True)


if __name__ == "__main__":
    main()
