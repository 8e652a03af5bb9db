import time, torch, sys
sys.path.insert(0, '.')
from oracle import densenet as odn
m = odn.DenseNet(num_classes=8, **odn.DENSENET121); m.eval()
x = torch.rand(256, 3, 128, 128)
for th in (16, 32, 64, 128):
    torch.set_num_threads(th)
    for chunk in (64, 256):
        with torch.no_grad():
            m(x[:chunk])
            t = time.time()
            for i in range(0, 256, chunk): m(x[i:i + chunk])
            dt = time.time() - t
        print('threads', th, 'chunk', chunk, '%.1f spots/s' % (256 / dt), flush=True)
