import numpy as np, sys
a=np.fromfile(sys.argv[1],dtype=np.uint64).reshape(-1,16).astype(np.int64)
a=a[a[:,0]>0]
t0=a[:,0].min()
print("tasks",len(a),"span us",(a[:,11].max()-t0)/100)
d=np.diff(a[:,:12],axis=1)/100.0
names=["A bitmaps","B eligible","CDE draws","F union","F flatten","report","G roots","scan","H draw","I flip","apply"]
for k,n in enumerate(names): print(f"{n:12s} mean {d[:,k].mean():7.2f} us  p90 {np.percentile(d[:,k],90):7.2f}")
print("total per task mean", (a[:,11]-a[:,0]).mean()/100, "start spread", np.percentile(a[:,0]-t0,[0,25,50,75,100])/100)

b=a
print("F: init %.2f  pack %.2f  barrier %.2f  unite %.2f" % (((b[:,12]-b[:,3]).mean()/100), ((b[:,13]-b[:,12]).mean()/100), ((b[:,14]-b[:,13]).mean()/100), ((b[:,4]-b[:,14]).mean()/100)))
n=b[:,15] & 0xFFFFFFFF; ov=b[:,15]>>32
print("list mean %.0f max %d overflow %.3f" % (n.mean(), n.max(), ov.mean()))
