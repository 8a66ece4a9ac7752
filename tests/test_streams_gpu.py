"""dfine_stream_fork (include/dfine_hip.h): the fork / join primitive of the side stream that carries the weight-gradient
launches - stream `to` must see everything enqueued on `from` before the call, across many more forks than the event ring
holds."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_stream_fork_orders_side_stream_after_main_and_back(cuda):
    from custom_d_fine_amd import hip
    side = torch.cuda.Stream(device=cuda)
    main = torch.cuda.current_stream(cuda)
    x = torch.zeros(1 << 24, device=cuda)                       # 64 MB: the fill takes long enough to be overtaken without a wait
    out = torch.empty(200, device=cuda)
    for i in range(200):                                        # > 64 forks each way: the ring wraps three times
        x.fill_(float(i + 1))                                   # main
        assert hip._lib.dfine_stream_fork(main.cuda_stream, side.cuda_stream) == 0
        with torch.cuda.stream(side):
            y = x[-1024:].sum() / 1024                          # must read the value just written by main
        assert hip._lib.dfine_stream_fork(side.cuda_stream, main.cuda_stream) == 0
        out[i] = y                                              # main, after the join
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), torch.arange(1, 201, dtype=torch.float32))


def test_side_stream_helpers_keep_inputs_alive_until_join(cuda):
    from custom_d_fine_amd import hip
    assert not hip._SIDE_LIVE
    st = hip._side_fork(cuda)
    assert st.cuda_stream != torch.cuda.current_stream(cuda).cuda_stream
    hip._SIDE_LIVE.append((torch.ones(4, device=cuda),))
    hip.side_join()
    assert not hip._SIDE_LIVE
    torch.cuda.synchronize()
