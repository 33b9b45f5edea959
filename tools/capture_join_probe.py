"""Round 6 probe: which cross-stream waits may a hipGraph capture contain on this runtime?  A = the capture's origin stream, B and C
side streams.  Variants (argv[1]): 0 B only; 1 C forked from / joined into A; 2 C also waits for an event of B (side <- side, one
direction); 3 as 2 and B then waits for an event of C (B <- C after C <- B); 4 as 3 with C entering the capture through B's event;
5 as 1 (control).  Measured (profiles/r06_capture_join_probe.txt): 0, 1, 2, 5 capture and replay; 3 and 4 die with a segmentation
fault inside capture_end.  => in CCTrainer a network's stream never waits for its auxiliary weight-gradient stream; the network's
tail continues ON the auxiliary stream and only the origin stream joins it."""
import sys, torch
dev = torch.device("cuda")
x = torch.ones(1 << 20, device=dev)
SB, SC = torch.cuda.Stream(), torch.cuda.Stream()
def body(v):
    A = torch.cuda.current_stream()
    B, C = SB, SC
    B.wait_stream(A)
    if v in (1, 2, 3, 5):
        C.wait_stream(A)
    with torch.cuda.stream(B):
        y = x * 2
    if v in (2, 3, 4):          # C waits on an event of B (side <- side)
        C.wait_stream(B)
    if v >= 1:
        with torch.cuda.stream(C):
            z = (y if v in (2, 3, 4) else x) + 1
    if v in (3, 4):             # B waits on an event of C (side <- side)
        B.wait_stream(C)
    with torch.cuda.stream(B):
        w = y * 1.5
    A.wait_stream(B)
    if v >= 1:
        A.wait_stream(C)
    return w
v = int(sys.argv[1])
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body(v)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    out = body(v)
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
print("variant", v, "ok", float(out[0]))
