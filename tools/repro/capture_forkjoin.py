"""Stream-capture fork/join patterns of the GP prior's deferred state update, with dummy kernels (ROCm 7.2 hipStreamEndCapture
crash hunt).  usage: capture_forkjoin.py <variant> ; prints OK <variant> when capture + 3 replays succeed."""
import sys
import torch

variant = sys.argv[1]
dev = torch.device("cuda:0")
bufs = [torch.zeros(1 << 16, device=dev) for _ in range(8)]
main = torch.cuda.Stream(device=dev)
sP, sA, sC, sK, sU = (torch.cuda.Stream(device=dev) for _ in range(5))
defer = "defer" in variant
early = "early" in variant
ahead = "ahead" in variant
viamain = "viamain" in variant       # chains A / C forked from main (behind evF) instead of from sP's event
own = "own" in variant               # state update on a stream of its own (sU)
prefork = "prefork" in variant
kmain = "kmain" in variant           # look-ahead stream forked from main behind an event of the state update's head
konc = "konc" in variant             # look-ahead work on chain C's stream (which entered the capture from main)


def op(i):
    bufs[i].add_(1.0)


def step():
    sP.wait_stream(main)
    if own:
        sP.wait_stream(sU)
    if ahead:
        sP.wait_stream(sC if konc else sK)
    with torch.cuda.stream(sP):
        op(0)
    op(1)
    ev_enc = torch.cuda.Event()
    ev_enc.record(main)
    op(2)
    if early:
        sP.wait_event(ev_enc)
        with torch.cuda.stream(sP):
            op(0)
        evF = torch.cuda.Event()
        evF.record(sP)
        main.wait_event(evF)
        if viamain:
            sA.wait_stream(main)
            sC.wait_stream(main)
        else:
            sA.wait_event(evF)
            sC.wait_event(evF)
    else:
        main.wait_stream(sP)
        op(0)
        sA.wait_stream(main)
        sC.wait_stream(main)
    with torch.cuda.stream(sC):
        op(3)
    with torch.cuda.stream(sA):
        op(4)
    op(1)
    if defer:
        sX = sU if own else sP
        if own:
            sX.wait_stream(sP)
        sX.wait_stream(sA)
        sX.wait_stream(sC)
        with torch.cuda.stream(sX):
            op(5)
            if ahead and kmain:
                e = torch.cuda.Event()
                e.record(sX)
                main.wait_event(e)
                sK.wait_stream(main)
                with torch.cuda.stream(sK):
                    op(6)
            elif ahead and konc:
                e = torch.cuda.Event()
                e.record(sX)
                sC.wait_event(e)
                with torch.cuda.stream(sC):
                    op(6)
            elif ahead:
                sK.wait_stream(sX)
                with torch.cuda.stream(sK):
                    op(6)
            op(5)
    else:
        main.wait_stream(sA)
        main.wait_stream(sC)
        op(5)
        if ahead:
            sK.wait_stream(main)
            with torch.cuda.stream(sK):
                op(6)
        op(5)


with torch.cuda.stream(main):
    step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main):
        if prefork:
            for s_ in (sP, sA, sC, sK, sU):
                s_.wait_stream(main)
        step()
        step()
        if defer:
            main.wait_stream(sU if own else sP)
        if ahead:
            main.wait_stream(sC if konc else sK)
        if prefork:
            for s_ in (sP, sA, sC, sK, sU):
                main.wait_stream(s_)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
print("OK", variant, [float(b[0]) for b in bufs[:7]])
