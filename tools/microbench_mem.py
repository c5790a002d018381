"""Memory-system probes on the MI355X (torch only): HBM copy ceiling and the
bandwidth of copies whose footprint fits the 256 MiB Infinity Cache.  Used to
decide how the two-pass facet transform should stage its intermediate."""
import json
import sys

import torch


def bw_copy(nbytes, iters):
    n = nbytes // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2 * nbytes / ms / 1e6  # GB/s (read + write)


def bw_pingpong(nbytes, iters):
    """a -> b then b -> a: every byte read was written by the previous kernel."""
    n = nbytes // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
        a.copy_(b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        b.copy_(a)
        a.copy_(b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (2 * iters)
    return 2 * nbytes / ms / 1e6


def main():
    out = {"device": torch.cuda.get_device_name(0), "copy": {}, "pingpong": {}}
    props = torch.cuda.get_device_properties(0)
    out["cus"] = props.multi_processor_count
    out["mem_gb"] = props.total_memory / 2**30
    for mb in (8, 16, 32, 64, 96, 128, 192, 256, 512, 1024, 4096):
        nb = mb << 20
        it = max(5, min(200, (8 << 30) // nb))
        out["copy"][mb] = round(bw_copy(nb, it), 1)
        out["pingpong"][mb] = round(bw_pingpong(nb, it), 1)
        print(mb, "MiB  copy GB/s", out["copy"][mb], " pingpong GB/s", out["pingpong"][mb], flush=True)
    json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/microbench_mem.json", "w"), indent=1)


if __name__ == "__main__":
    main()
