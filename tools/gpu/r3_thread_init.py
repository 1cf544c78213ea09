"""What does the first HIP work on a new host thread cost?  (explains the fixed cost of the threaded feeder on short clips)"""
import os, sys, threading, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd"))
import numpy as np
from svsdct import native
from svsdct.pipeline import FramePipeline
native.ensure_device(0)
pipe = FramePipeline(1080, 1920, 8, 8, 3, depth=3, mode="guarded")
pipe.set_payload(np.zeros(1000, np.uint8))
pipe.submit_embed(0, 8, 0); pipe.embed_result(0)
t = {}
def work(tag):
    t0 = time.perf_counter(); pipe.bind_thread(); t1 = time.perf_counter()
    pipe.submit_embed(1, 8, 0); t2 = time.perf_counter()
    pipe.embed_result(1); t3 = time.perf_counter()
    pipe.submit_embed(1, 8, 0); pipe.embed_result(1); t4 = time.perf_counter()
    t[tag] = (t1 - t0, t2 - t1, t3 - t2, t4 - t3)
work("main thread")
for i in range(3):
    th = threading.Thread(target=work, args=(f"new thread {i}",)); th.start(); th.join()
for k, v in t.items():
    print(f"{k:14s} bind {v[0]*1e3:7.2f} ms  first submit {v[1]*1e3:7.2f} ms  wait {v[2]*1e3:7.2f} ms  second submit+wait {v[3]*1e3:7.2f} ms")
pipe.close()
