"""Grid-of-images figure (/root/reference/image_grid.py:6-29): ``imgs`` is [rows, cols, H, W, C]."""
import os

import numpy as np


def create_image_grid(imgs, figsize=None, cmap='gray'):
    import matplotlib
    matplotlib.use("Agg", force=False)
    from matplotlib import pyplot as plt, gridspec
    rows, cols = imgs.shape[0], imgs.shape[1]
    fig = plt.figure(figsize=figsize if figsize is not None else (rows, cols))
    spec = gridspec.GridSpec(rows, cols)
    spec.update(wspace=0.025, hspace=0.025)
    for r in range(rows):
        for c in range(cols):
            ax = fig.add_subplot(spec[r, c])
            cell = np.asarray(imgs[r, c])
            ax.imshow(cell[:, :, 0] if cell.ndim == 3 and cell.shape[2] == 1 else cell, cmap=cmap)
            ax.axis('off')
    return fig


def write_image_grid(filepath, imgs, figsize=None, cmap='gray'):
    from matplotlib import pyplot as plt
    os.makedirs(os.path.dirname(os.path.abspath(filepath)), exist_ok=True)
    fig = create_image_grid(imgs, figsize, cmap=cmap)
    fig.savefig(filepath)
    plt.close(fig)
