import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
t0=time.perf_counter()
import torch
t1=time.perf_counter()
torch.cuda.init(); x=torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t2=time.perf_counter()
import numpy as np
y=torch.from_numpy(np.zeros(65536,np.float32)).cuda(); torch.cuda.synchronize()
t3=time.perf_counter()
from sushi_amd import _native
L=_native.lib()
t4=time.perf_counter()
from sushi_amd.device import DeviceStream
d=DeviceStream(np.linspace(0,1,65536,dtype=np.float32), _wait_for_warm_up=False); torch.cuda.synchronize()
t5=time.perf_counter()
d.searchable(); torch.cuda.synchronize()
t6=time.perf_counter()
L.sushi_hip_device_prepare(); torch.cuda.synchronize()
t7=time.perf_counter()
print({"import_torch_ms": round((t1-t0)*1e3,1), "cuda_init_first_alloc_ms": round((t2-t1)*1e3,1), "first_h2d_ms": round((t3-t2)*1e3,1), "load_lib_ms": round((t4-t3)*1e3,1), "first_stream_prep_ms": round((t5-t4)*1e3,1), "first_spectra_ms": round((t6-t5)*1e3,1), "device_prepare_ms": round((t7-t6)*1e3,1)})
