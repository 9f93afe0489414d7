"""CPU: chunk planning of one long sequence over pipeline streams / ranks (sindslam_amd/sequence.py)."""
import pytest

from sindslam_amd.sequence import plan_chunks


@pytest.mark.parametrize("n,chunks,warm", [(4000, 64, 5), (823, 8, 5), (10, 4, 3), (3, 8, 5), (100, 1, 5), (17, 16, 0)])
def test_chunks_partition_the_sequence(n, chunks, warm):
    cs = plan_chunks(n, chunks, warm)
    assert len(cs) == chunks and cs[0].first == 1 and cs[0].start == 1 and cs[-1].last == n
    for a, b in zip(cs, cs[1:]):
        assert a.last == b.first                              # contiguous, no frame twice
    sizes = [c.last - c.first for c in cs]
    assert sum(sizes) == n - 1 and max(sizes) - min(sizes) <= 1
    for c in cs[1:]:
        assert c.start == max(1, c.first - warm) and c.processed == c.last - c.start


def test_ranks_take_disjoint_blocks_of_chunks():
    cs = plan_chunks(4000, 8 * 64, 5)                          # 8 ranks x 64 streams
    owned = [set(range(c.first, c.last)) for c in cs]
    for r in range(8):
        mine = set().union(*owned[r * 64:(r + 1) * 64])
        assert 64 * 7 <= len(mine) <= 64 * 8                   # 3999 frames in 512 chunks of 7 or 8: every stream of every rank runs the same number of steps
    assert set().union(*owned) == set(range(1, 4000))


def test_bad_arguments():
    with pytest.raises(ValueError):
        plan_chunks(1, 4)
    with pytest.raises(ValueError):
        plan_chunks(10, 0)
