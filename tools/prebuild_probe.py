import sys, os, glob, subprocess, time
from boda_amd import rtc
from boda_amd.op import Op, Dims, Nda
def conv_op(B,C,H,W,OC,K=1,S=1,P=0):
    OH=(H+2*P-K)//S+1; OW=(W+2*P-K)//S+1
    d=lambda n,s: Nda(Dims(n,s,"float"))
    none=lambda yx: Nda(Dims(("y","x"),tuple(yx),"none"),"none")
    return Op({"type":"Convolution","func_name":"hip_conv"},{"in":d(("img","chan","y","x"),(B,C,H,W)),"filts":d(("out_chan","in_chan","y","x"),(OC,C,K,K)),
        "biases":d(("out_chan",),(OC,)),"out":d(("img","chan","y","x"),(B,OC,OH,OW)),"stride":none((S,S)),"in_pad":none((P,P)),"kern_sz":none((K,K)),
        "conv_has_relu":Nda(None,"uint32_t",(1,))})
t0=time.time()
a=[int(x) for x in sys.argv[1:8]]+[1,1,0][len(sys.argv)-5:] if len(sys.argv)<8 else [int(x) for x in sys.argv[1:8]]
B,C,H,OC,K,S,P=a
op=conv_op(B,C,H,H,OC,K,S,P)
if os.environ.get("BF16"): op.str_vals["func_name"]="hip_conv_bf16"
n=rtc.prebuild(op)
print("code bytes",n)
f=max(glob.glob("boda_amd/_kcache/*.hsaco"),key=os.path.getmtime)
print(f)
out=subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf","--notes",f],capture_output=True,text=True).stdout
for l in out.splitlines():
    if any(k in l for k in ("vgpr_count","sgpr_count","spill","private_segment_fixed","group_segment_fixed","agpr",".name:")): print(l.strip())
