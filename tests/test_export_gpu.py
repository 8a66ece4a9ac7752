"""(f4) the export artefact: `torch.export` ExportedProgram of the deployed eval forward + post-processor with every C-ABI launch
as one `dfine::call` node (custom_d_fine_amd/dl/export_program.py; the reference's counterpart is `export_to_onnx`,
src/dl/export.py:131-173).  The saved program, loaded back, must reproduce the eager HIP forward and hold no library GEMM /
convolution operator of its own."""
import pytest
import torch

from custom_d_fine_amd.d_fine import dfine
from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size,img,batch,half", [("n", 320, 2, True), ("s", 320, 1, True), ("s", 320, 1, False)])
def test_exported_program_reproduces_eager_forward(cuda, tmp_path, size, img, batch, half):
    from custom_d_fine_amd.dl.export import DFINEPostProcessor, ExportWrapper
    from custom_d_fine_amd.dl.export_program import count_launch_nodes, export_program, graph_targets, load_program
    m = dfine.build_model(size, 80, False, "cpu", img_size=[img, img])
    m.load_state_dict(helpers.seeded_state_dict(m.state_dict()))
    m = m.to(cuda)
    path, ep = export_program(m, 80, (img, img), tmp_path / "model.pt2", batch=batch, half=half)
    assert path.exists() and path.stat().st_size > 1 << 20
    n_launch = count_launch_nodes(ep)
    targets = graph_targets(ep)
    assert n_launch > 100, n_launch
    heavy = [t for t in targets if any(k in t for k in ("convolution", "aten.mm", "aten.addmm", "aten.bmm", "aten.linear",
                                                        "scaled_dot_product", "native_batch_norm", "native_layer_norm"))]
    assert not heavy, sorted(set(heavy))
    x = helpers.make_images(batch, img).to(cuda)
    wrapper = ExportWrapper(m, DFINEPostProcessor(80), (img, img)).eval()        # (m was deployed by export_program)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=half, cache_enabled=False):
        want = wrapper(x)
    run = load_program(path)
    with torch.no_grad():
        got = run(x)
    assert len(got) == len(want) == 3
    assert torch.equal(got[0], want[0]), "labels differ"
    assert torch.allclose(got[1], want[1], rtol=0, atol=1e-3) and torch.allclose(got[2], want[2], rtol=0, atol=1e-5)
    # another input through the same program: nothing of the example input was baked in
    x2 = helpers.make_images(batch, img, seed=9).to(cuda)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=half, cache_enabled=False):
        want2 = wrapper(x2)
    with torch.no_grad():
        got2 = run(x2)
    assert torch.equal(got2[0], want2[0]) and torch.allclose(got2[1], want2[1], rtol=0, atol=1e-3)
