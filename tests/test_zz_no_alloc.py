"""The entry point that allocates nothing (rt_compare_frames_noalloc) against rt_compare_frames: same count, same
difference image, on two calls in a row of which the second is smaller (the static scratch array keeps the first call's
count, so the second must clear it)."""
import numpy as np
import pytest

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu

rt = ge.load_package()


def _pair(rng, w, h, torch):
    a = rng.integers(0, 1 << 32, size=(h, w), dtype=np.uint32)
    b = a.copy()
    flat = b.reshape(-1)
    touched = rng.permutation(a.size)[: max(1, a.size // 3)]
    flat[touched] = rng.integers(0, 1 << 32, size=len(touched), dtype=np.uint32)
    nudged = rng.permutation(a.size)[: a.size // 5]                       # one step in the blue channel: tolerated by the comparison
    flat[nudged] = (flat[nudged] & 0xFFFFFF00) | np.minimum((flat[nudged] & 0xFF) + 1, 255)
    return (torch.from_numpy(a.view(np.int32)).to("cuda:0"), torch.from_numpy(b.view(np.int32)).to("cuda:0"))


def test_compare_without_allocation_equals_compare():
    import torch
    rt.init(0)
    rng = np.random.default_rng(77)
    count_dev = torch.full((1,), -1, dtype=torch.int64, device="cuda:0")
    seen = []
    for w, h in ((1921, 1081), (333, 7), (5, 1)):                          # big first, then smaller ones
        ta, tb = _pair(rng, w, h, torch)
        want_diff = torch.full((h, w), -1, dtype=torch.int32, device="cuda:0")
        rt.after_torch()                                                   # the fills run on torch's stream: the library's lane waits for them (an event, no wait on the host)
        want = rt.compare_frames(ta.data_ptr(), (w, h), tb.data_ptr(), (w, h), want_diff.data_ptr())
        got_diff = torch.full((h, w), -2, dtype=torch.int32, device="cuda:0")
        rt.after_torch()
        rt.compare_frames_noalloc(ta.data_ptr(), (w, h), tb.data_ptr(), (w, h), count_dev.data_ptr(), got_diff.data_ptr())
        rt.synchronize()
        assert int(count_dev.item()) == want and want > 0
        assert torch.equal(got_diff, want_diff)
        rt.compare_frames_noalloc(ta.data_ptr(), (w, h), ta.data_ptr(), (w, h), count_dev.data_ptr())      # equal frames, no difference image
        rt.synchronize()
        assert int(count_dev.item()) == 0
        seen.append(want)
    assert seen[0] > seen[1] > 0
    with pytest.raises(rt.RtError, match="Images have different dimensions"):
        rt.compare_frames_noalloc(ta.data_ptr(), (5, 1), tb.data_ptr(), (1, 5), count_dev.data_ptr())
    with pytest.raises(rt.RtError, match="NULL"):
        rt.compare_frames_noalloc(ta.data_ptr(), (5, 1), tb.data_ptr(), (5, 1), 0)
    with pytest.raises(rt.RtError, match="2\\^31 pixels"):
        rt.compare_frames_noalloc(ta.data_ptr(), (65536, 65536), tb.data_ptr(), (65536, 65536), count_dev.data_ptr())
