import sys, time, numpy as np
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'rtl-ws_amd'))
import rtlws
from rtlws import synth
eng=rtlws.Engine(0); L=rtlws.hip_lib()
for N,K,rows in ((1024,128,1),(1024,6,1),(1024,1,1),(4096,1,1),(8192,1,1),(1024,1,256)):
    iq=synth.tone_noise_iq(K*rows,N,seed=1)
    d_in=eng.upload(iq); d_out=eng.alloc(rows*N*8)
    desc=rtlws.make_desc(N,k_avg=K)
    for _ in range(3): eng.spectra_batch_f64(desc,d_in,K*rows,d_out)
    eng.sync()
    e0,e1=L.rtlws_event_create(),L.rtlws_event_create()
    L.rtlws_event_record(e0,eng.h,None)
    for _ in range(50): eng.spectra_batch_f64(desc,d_in,K*rows,d_out)
    L.rtlws_event_record(e1,eng.h,None)
    print("f64 N=%d K=%d rows=%d: %.1f us per launch" % (N,K,rows,1e3*L.rtlws_event_elapsed_ms(e0,e1)/50))
