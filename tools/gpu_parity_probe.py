import sys, os, time, importlib, ctypes as C
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np
import oracle_py as op
capi = op.capi
rtx = importlib.import_module("ray-tracing_amd")
scenes = importlib.import_module("ray-tracing_amd.scenes")
print("devices", rtx.device_count())
def cmp(cfg,W,H,N,seed=1):
    ob=op.OracleBuilder(10); d=scenes.assemble(ob,cfg)
    p=capi.default_params(W,H,N,seed)
    t=time.time(); rgb1,f1,h1,st=op.render(d,p); t1=time.time()-t
    sc=rtx.Scene(d)
    t=time.time(); rgb2,h2=sc.render_hits(p); t2=time.time()-t
    p2=capi.default_params(W,H,N,seed)
    rgb3,gst=sc.render(p2,timing=True)
    same_ids = len(h1)==len(h2) and all(np.array_equal(h1[k],h2[k]) for k in ("pixel","ordinal","leaf","prim"))
    tbits = len(h1)==len(h2) and np.array_equal(h1["t"].view(np.uint32),h2["t"].view(np.uint32))
    diff=np.abs(rgb1.astype(int)-rgb2.astype(int)).max(axis=2)
    cdiff = np.abs(h1["color"]-h2["color"]).max() if len(h1)==len(h2) and len(h1) else -1
    crel = (np.abs(h1["color"]-h2["color"])/(np.abs(h1["color"])+1e-3)).max() if len(h1)==len(h2) and len(h1) else -1
    print(cfg,W,H,N,"cpu %.2fs gpu(hits) %.2fs"%(t1,t2),"hits",len(h1),len(h2),"ids eq",same_ids,"t bits eq",tbits,
      "rgb exact %.4f%% within1 %.4f%%"%(100*(diff==0).mean(),100*(diff<=1).mean()),"maxdiff",diff.max(),"rec color maxabs %.3g maxrel %.3g"%(cdiff,crel), "rgb3==rgb2",np.array_equal(rgb2,rgb3))
    print("   kernel ms", list(gst.kernel_ms), "hits",gst.hits,"shadow tests",gst.shadow_tests, "oracle shadow tests", st.shadow_prim_tests)
cmp("spheres",200,200,4)
cmp("default",1000,1000,1)
cmp("default",250,250,8)
cmp("multi",384,216,70,seed=2)
cmp("bunny",1920,1080,100)


def ieee_vs_fast(cfg, W, H, N, seed=1, depth=0):
    ob = op.OracleBuilder(10); d = scenes.assemble(ob, cfg)
    sc = rtx.Scene(d)
    p = capi.default_params(W, H, N, seed); p.max_depth = depth
    rgb_f, h_f = sc.render_hits(p)
    p2 = capi.default_params(W, H, N, seed); p2.max_depth = depth; p2.flags |= capi.RTX_FLAG_IEEE_SHADE
    rgb_i, h_i = sc.render_hits(p2)
    rel = (np.abs(h_f["color"] - h_i["color"]) / (np.abs(h_i["color"]) + 1e-2)).max()
    diff = np.abs(rgb_f.astype(int) - rgb_i.astype(int)).max(axis=2)
    print("ieee vs fast", cfg, W, H, N, "depth", depth, "max rel colour diff %.3g" % rel,
          "rgb equal %.5f%% maxdiff %d" % (100 * (diff == 0).mean(), diff.max()))


ieee_vs_fast("spheres", 500, 500, 64)
ieee_vs_fast("dino_spheres", 960, 540, 200)
ieee_vs_fast("multi", 960, 540, 300)
ieee_vs_fast("spheres", 300, 300, 40, depth=2)
