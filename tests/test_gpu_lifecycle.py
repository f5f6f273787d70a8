"""Renderer lifetime on the GPU: pt_destroy gives back everything a renderer held (the BVH builder's kept scratch was only
freed by release_all(), which nothing called).  Renderer::~Renderer (renderer_pt.hpp:34) releases its Metal objects through
NS::SharedPtr; here every device array is a DevBuf or explicitly released in pt_renderer::release_all()."""
import pytest

from platinum_amd import Renderer, scenes


@pytest.mark.gpu
def test_destroy_returns_all_device_memory_including_the_bvh_builder_scratch():
    import torch
    sc = scenes.field_scene(16)          # 259 k triangles: ~0.1 GB of builder scratch, ~70 MB of structure
    free = []
    for i in range(4):
        r = Renderer(device=0)
        r.startRender(sc, (256, 144), 4, max_bounces=4)
        r.render(0)
        r.readbackAccumulator()
        if i == 1:                       # a restart on the same renderer must not grow it either
            r.startRender(sc, (256, 144), 4, max_bounces=4)
            r.render(0)
            r.wait()
        r.close()
        torch.cuda.synchronize()
        free.append(torch.cuda.mem_get_info(0)[0])
    # the first create may leave runtime-internal pools behind (code objects, signal pools); after that free memory must not drift
    assert abs(free[3] - free[1]) <= 8 << 20, [f / 2**20 for f in free]
    assert abs(free[2] - free[1]) <= 8 << 20, [f / 2**20 for f in free]
