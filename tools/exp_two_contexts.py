# Do two independent renders share the GPU better than one after the other? Two contexts, two host threads.
import sys, time, threading
sys.path.insert(0, '.')
import path_tracer_amd as pt
sc = pt.LoadedScene("cornell-box-guided")
ctxs = [pt.Context(0), pt.Context(0)]
for c in ctxs:
    c.load(sc); c.render(sc.integrator, 64)
t = time.time(); ctxs[0].render(sc.integrator, 512); one = time.time() - t
t = time.time(); ctxs[0].render(sc.integrator, 512); one = min(one, time.time() - t)
print(f"one render alone: {one:.3f} s", flush=True)
for rep in range(2):
    th = [threading.Thread(target=lambda c=c: c.render(sc.integrator, 512)) for c in ctxs]
    t = time.time()
    for x in th: x.start()
    for x in th: x.join()
    both = time.time() - t
    print(f"two renders together: {both:.3f} s = {both / 2:.3f} s each -> {2 * one / both:.3f}x the serial rate", flush=True)
