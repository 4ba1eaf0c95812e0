"""FMX_BUILD_TRACE of the config-4 builds (n = 2^30 bytes, sigma = 255): FM and RLFM, twice each"""
import sys, os, time
# (FMX_BUILD_TRACE is read by the measurement build only since round 6: FMX_LIB=fm_index_amd/libfmx_measure.so)
os.environ.setdefault("FMX_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "fm_index_amd", "libfmx_measure.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fm_index_amd as F
from fm_index_amd import workload as W
dev = torch.device("cuda", 0)
N = 1 << 30
text = W.byte_text_torch(N, 17, dev)
for cls in (F.FMIndexWithLocate, F.RLFMIndexWithLocate):
    for rep in range(2):
        sys.stderr.write("== %s rep %d\n" % (cls.__name__, rep)); sys.stderr.flush()
        t0 = time.time()
        ix = cls.from_device_text(text.data_ptr(), N, 255, level=2)
        sys.stderr.write("wall %.3f s\n" % (time.time() - t0))
        ix.close()
