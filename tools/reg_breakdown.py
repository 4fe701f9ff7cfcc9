import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import voronoirt_amd as vrt
dev = torch.device('cuda', 0)
nz, nx, ny = 215, 130, 130
z = np.linspace(-0.5e6, 14e6, nz); x = np.linspace(0, 6e6, nx); y = np.linspace(0, 6e6, ny)
w, th, ph, nq = vrt.read_quadrature('ul7n12.dat'); ks = vrt.quadrature_directions(th, ph)
nlam = 8; ns = nq * nlam
g = torch.Generator(device=dev); g.manual_seed(1)
S = 1 + 0.1 * torch.rand((nlam, ny, nx, nz), generator=g, device=dev, dtype=torch.float64)
al = 1e-5 * (1 + torch.rand((nlam, ny, nx, nz), generator=g, device=dev, dtype=torch.float64))
I0 = torch.rand((ns, ny, nx), generator=g, device=dev, dtype=torch.float64)
out = torch.empty((ns, ny, nx, nz), device=dev, dtype=torch.float64)
sol = vrt.RegularSolver(z, x, y)
vol = nz * nx * ny
for sweeps in (1, 2, 3):
    for _ in range(2):
        sol.execute_dev(np.repeat(ks, nlam, axis=0), np.repeat(th > 90, nlam), S.data_ptr(), vol, al.data_ptr(), vol,
                        I0.data_ptr(), out.data_ptr(), sweeps, torch.cuda.current_stream().cuda_stream, field_period=nlam)
        torch.cuda.synchronize()
    print('n_sweeps', sweeps, 'solve kernel ms', sol.last_solve_ms())
# per direction alone (1 solve each)
for a in range(nq):
    sol.execute_dev(ks[a:a+1], [th[a] > 90], S.data_ptr(), 0, al.data_ptr(), 0, I0.data_ptr(), out.data_ptr(), 3,
                    torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print('direction', a, 'theta %.1f phi %.1f' % (th[a], ph[a]), 'ms', round(sol.last_solve_ms(), 2))
