"""Host-side generation of the separable 1-D Sobolev filter (reference:
nonrigid_opt/slavcheva/sobolev_filter.py:208-252).  A one-off 343x343 solve + SVD: stays on the host."""
import numpy as np


def stencil_laplacian_matrix(size, precision=np.float32):
    """7-point Laplacian on the size^3 block in flat (x fastest) ordering.  Neighbours are accepted by a test on
    the FLAT index only (sobolev_filter.py:154-158), so x/y neighbours wrap across rows at the block faces --
    the reference's behaviour, kept."""
    n = size ** 3
    lap = np.zeros((n, n), precision)
    offsets = (-1, 1, size, -size, -size * size, size * size)
    for i in range(n):
        lap[i, i] = -6.0
        for off in offsets:
            j = i + off
            if 0 <= j < n:
                lap[i, j] = 1.0
    return lap


def generate_1d_sobolev_kernel(size=7, strength=0.1, precision=np.float32):
    """solve (I - strength*L) S = e_centre on the size^3 grid and return the dominant left singular vector of the
    mode-1 unfolding of S, signed so that the centre tap is positive."""
    n = size ** 3
    rhs = np.zeros((n, 1), precision)
    rhs[n // 2] = 1.0
    system = np.identity(n, precision) - strength * stencil_laplacian_matrix(size, precision)
    cube = np.linalg.solve(system, rhs).reshape((size, size, size))
    unfolded = np.moveaxis(cube, 1, 0).reshape(size, -1)
    u, _, _ = np.linalg.svd(unfolded)
    kernel = u[:, 0]
    if kernel[size // 2] < 0:
        kernel = -kernel
    return kernel.astype(precision)
