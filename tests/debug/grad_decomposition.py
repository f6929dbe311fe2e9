"""Diagnostic (not a test, CPU only): splits the f32 gradient error of ill-conditioned soak draws into the part made by the sums over
pixels (render-backward) and the part made by the per-splat algebra (preprocess-backward): python tests/debug/grad_decomposition.py
(output kept in profiles/r04_gradient_error_survey.txt)"""
import sys, ctypes as C, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from oracle import Oracle, _ptr
from test_gpu_random_sweep import _draw
o32,o64=Oracle("f32"),Oracle("f64")
def stages(o, scene, pose, W,H,fov,bg,sm,dL):
    cam=o.lookat(*pose,width=W,height=H,fov=fov)
    P=scene["pos"].shape[0]
    color=o.sh_process(np.array(cam.position[:]), scene["pos"], scene["sh"])
    m,d,c=o.project(scene["pos"],scene["scale"],scene["rotq"],cam,scale_modifier=sm)
    mp,conic,tiles,rad=o.allocate_tiles(W,H,d,m,c)
    offs=o.inclusive_sum(tiles)
    k,v=o.copy_with_keys(W,H,mp,offs,rad,d)
    ks,vs=o.sort_pairs(k,v)
    G=((W+15)//16)*((H+15)//16)
    rng=o.get_ranges(ks,G)
    img,fT,nc,_=o.render_forward(W,H,bg,rng,vs,mp,conic,scene["opacity"],color)
    gm=np.zeros((P,2),o.dtype); gc=np.zeros((P,3),o.dtype); go=np.zeros(P,o.dtype); gcol=np.zeros((P,3),o.dtype)
    dLa=o.arr(dL); bga=o.arr(bg); op=o.arr(scene["opacity"])
    o.lib.orc_render_backward(C.c_int(W),C.c_int(H),o.rp(bga),_ptr(np.ascontiguousarray(rng,dtype=np.uint32),C.c_uint32),_ptr(np.ascontiguousarray(vs,dtype=np.uint32),C.c_uint32),
        o.rp(o.arr(mp)),o.rp(o.arr(conic)),o.rp(op),o.rp(o.arr(color)),o.rp(o.arr(fT)),_ptr(nc,C.c_uint32),o.rp(dLa),o.rp(gm),o.rp(gc),o.rp(go),o.rp(gcol))
    return cam,rad,gm,gc,gcol
rel=lambda a,b: np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-30)
for seed in (2124,2412,1584,2352):
    rng,scene,W,H,pose,fov,bg,sm=_draw(seed)
    dL=rng.normal(size=(3,H,W)).astype(np.float32)
    c32,r32,gm32,gc32,gcol32=stages(o32,scene,pose,W,H,fov,bg,sm,dL)
    c64,r64,gm64,gc64,gcol64=stages(o64,scene,pose,W,H,fov,bg,sm,dL)
    full64=o64.preprocess_backward(scene,c64,r64,gm64,gc64,gcol64,scale_modifier=sm)
    full32=o32.preprocess_backward(scene,c32,r32,gm32,gc32,gcol32,scale_modifier=sm)
    a=o32.preprocess_backward(scene,c32,r32,gm64.astype(np.float32),gc64.astype(np.float32),gcol64.astype(np.float32),scale_modifier=sm)   # f32 algebra, exact sums
    b=o64.preprocess_backward(scene,c64,r64,gm32,gc32,gcol32,scale_modifier=sm)   # exact algebra, f32 sums
    print("seed",seed,"2D grads f32 vs f64: mean %.1e conic %.1e color %.1e"%(rel(gm32,gm64),rel(gc32,gc64),rel(gcol32,gcol64)))
    for k in ("pos","scale","rotq"):
        print("  %-6s full f32 %.2e | f32 algebra on exact 2-D grads %.2e | exact algebra on f32 2-D grads %.2e"%(k,rel(full32[k],full64[k]),rel(a[k],full64[k]),rel(b[k],full64[k])))
