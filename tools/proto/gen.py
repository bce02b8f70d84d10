import numpy as np, torch, math
def eight_gaussians(n, seed):
    g = torch.Generator().manual_seed(seed)
    centers = torch.tensor([(1,0),(-1,0),(0,1),(0,-1),(1/math.sqrt(2),1/math.sqrt(2)),(1/math.sqrt(2),-1/math.sqrt(2)),(-1/math.sqrt(2),1/math.sqrt(2)),(-1/math.sqrt(2),-1/math.sqrt(2))],dtype=torch.float32)*5
    noise = torch.randn(n,2,generator=g)*math.sqrt(math.sqrt(0.1))
    idx = torch.randint(0,8,(n,),generator=g)
    return (centers[idx]+noise).float()
def moons(n, seed):
    rng = np.random.RandomState(seed)
    no = n//2; ni = n-no
    to = np.linspace(0,np.pi,no); ti=np.linspace(0,np.pi,ni)
    X = np.vstack([np.c_[np.cos(to),np.sin(to)], np.c_[1-np.cos(ti),0.5-np.sin(ti)]])
    X += rng.rand(n,1)*0.2
    return torch.from_numpy(X.astype(np.float32))*3-1
def mnist_like(n, seed):
    g = torch.Generator().manual_seed(seed)
    mu = torch.rand(10,784,generator=g)*2-1
    k = torch.randint(0,10,(n,),generator=g)
    return torch.clamp(0.35*torch.randn(n,784,generator=g)+mu[k],-1,1)
def cost(x0,x1):
    a=x0.double().numpy(); b=x1.double().numpy()
    M=((a*a).sum(1)[:,None]+(b*b).sum(1)[None,:]-2*a@b.T)
    # direct for accuracy
    if a.shape[1]<=8:
        M=((a[:,None,:]-b[None,:,:])**2).sum(-1)
    return np.maximum(M,0).astype(np.float32)
def get(cfg,n=4096):
    if cfg=='d2': return eight_gaussians(n,0), moons(n,0)
    if cfg=='d784':
        g=torch.Generator().manual_seed(1000); return torch.randn(n,784,generator=g), mnist_like(n,1001)
