cd $GRAFT_REPO_ROOT
timeout -k 10 120 python - <<'PY'
import sys, torch, numpy as np
sys.path[:0]=['.','tests']
from raytracer_amd.scene import Scene
from raytracer_amd import abi
dev=torch.device('cuda',0)
for (w,h,spp,d,don) in [(640,480,1,5,1),(3840,2160,4,5,1)]:
    s=Scene(width=w,height=h,samplesPerPixel=spp,maxTraceDepth=d,jitter=0,seed=1,work_donation=don)
    rgba=torch.zeros((w,h,4),dtype=torch.float32,device=dev); depth=torch.zeros((w,h),dtype=torch.float32,device=dev)
    st=torch.zeros(abi.RT_STATS_WORDS,dtype=torch.int64,device=dev)
    for rep in range(2):
        s.renderColumnsDevice(rgba.data_ptr(),depth.data_ptr(),None,st.data_ptr(),torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    v=st.cpu().numpy()
    it=v[2]
    print(w,h,'rays',v[0],'waves',v[7],'iters',it,'clk/iter per wave',v[6]/it,'pop',v[8]/it,'prim',v[9]/it,'childgen',v[12]/it,'scan',v[13]/it,'trav',v[14]/it,'shade',v[15]/it, flush=True)
PY
