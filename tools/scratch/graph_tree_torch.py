"""hipGraph capture of a tree of streams with torch alone (no library kernels): which shape of fork / join makes
hipStreamEndCapture crash.  usage: python3 graph_tree_torch.py --variant chain|two|tree|tree_norecord"""
import argparse
import ctypes
import faulthandler
import os

import torch

if not os.environ.get("SEGV_BT"):
    faulthandler.enable()
if os.environ.get("SEGV_BT"):
    ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "segv_bt.so")).install_segv_bt()


def pair(main, side, fa, fb, record=True):
    side.wait_stream(main)
    with torch.cuda.stream(side):
        b = fb()
    a = fa()
    main.wait_stream(side)
    if record:
        b.record_stream(main)
    return a, b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="tree")
    v = ap.parse_args().variant
    x = torch.arange(1 << 20, device="cuda", dtype=torch.int64)
    s = [torch.cuda.Stream() for _ in range(3)]
    rec = not v.endswith("_norecord")

    def circuit():
        m = torch.cuda.current_stream()
        if v.startswith("chain"):        # m forks s0, s0 forks s1
            def fb():
                p, q = pair(s[0], s[1], lambda: x + 3, lambda: x + 4, rec)
                return p + q
            a, b = pair(m, s[0], lambda: x + 1, fb, rec)
            return a + b
        if v == "two":          # m forks s0, joins, forks s1, joins
            a, b = pair(m, s[0], lambda: x + 1, lambda: x + 2)
            c, d = pair(m, s[1], lambda: x + 3, lambda: x + 4)
            return a + b + c + d

        def fa():               # tree: m forks s0; m forks s1; s0 forks s2
            p, q = pair(m, s[1], lambda: x + 1, lambda: x + 2, rec)
            return p + q

        def fb():
            p, q = pair(s[0], s[2], lambda: x + 3, lambda: x + 4, rec)
            return p + q
        a, b = pair(m, s[0], fa, fb, rec)
        return a + b

    want = circuit()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        circuit()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print("warm ok", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = circuit()
    print("capture ok", flush=True)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    print("replay equal ok", flush=True)


if __name__ == "__main__":
    main()
