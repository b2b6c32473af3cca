"""The one reference-held pin of this path: tests/golden/rooms_emb.npz is written by tests/golden/make_rooms_golden.py from
the reference's own rooms.py (UTSRoom.return_embedding, rooms.py:46-99).  It pins the `emb` input contract: 16 integers per
position, values in [26, 1281] - inside Embedding(2000, 256) (dl_models/u_net.py:257) - and, value for value, the
product's table-driven restatement (unet_rir_amd/rooms.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "rooms_emb.npz"))


def test_fixture_known_answer():
    """['LargeMeetingRoom','B','Circular','22','1'] -> this vector; 16 ints, global range [26, 1281]."""
    assert (GOLD["sample_room"][0], GOLD["sample_zone"][0], GOLD["sample_array"][0], int(GOLD["sample_speaker"][0]),
            int(GOLD["sample_mic"][0])) == ("LargeMeetingRoom", "B", "Circular", 22, 1)
    assert GOLD["sample_emb"][0].tolist() == [994, 923, 1087, 1022, 81, 105, 81, 92, 300, 380, 392, 145, 537, 498, 145, 1281]
    assert GOLD["sample_emb"].shape[1] == 16
    assert (int(GOLD["global_min"]), int(GOLD["global_max"])) == (26, 1281)


def test_product_restatement_matches_reference_values():
    from unet_rir_amd import rooms
    for i in range(len(GOLD["sample_room"])):
        got = rooms.uts_room_embedding(str(GOLD["sample_room"][i]), str(GOLD["sample_zone"][i]), str(GOLD["sample_array"][i]),
                                       int(GOLD["sample_speaker"][i]), int(GOLD["sample_mic"][i]))
        assert got == GOLD["sample_emb"][i].tolist(), i
    # every position of every room: per-room extremes equal the reference's
    names = [str(n) for n in GOLD["room_names"]]
    n = 0
    for k, name in enumerate(names):
        lo, hi = 10 ** 9, -10 ** 9
        for z in rooms.ZONE_OFFSET:
            for arr, n_mic in (("Planar", 64), ("Circular", 60)):
                for l in range(1, 61):
                    for m in range(1, n_mic + 1):
                        v = rooms.uts_room_embedding(name, z, arr, l, m)
                        lo, hi = min(lo, min(v)), max(hi, max(v))
                        n += 1
        assert (lo, hi) == (int(GOLD["room_min"][k]), int(GOLD["room_max"][k])), name
        args = GOLD["room_args"][k]
        a = rooms.UTS_ROOMS[name]
        assert list(a[:9]) + list(a[9]) + [a[10]] == args.tolist(), name
    assert n == int(GOLD["n_combinations"])
    assert (rooms.EMB_MIN, rooms.EMB_MAX) == (int(GOLD["global_min"]), int(GOLD["global_max"]))
    pair = rooms.emb_pair(("LargeMeetingRoom", "B", "Circular", 22, 1), ("SmallMeetingRoom", "E", "Planar", 60, 64))
    assert pair.shape == (2, 16) and pair.dtype == np.int32 and pair[1].tolist() == GOLD["sample_emb"][2].tolist()


def test_synthetic_generators_and_embedding_bounds_agree_with_the_fixture():
    """The synthetic `emb` of every generator in the build draws from exactly the fixture's range, and the embedding table
    (VOCAB rows) covers it."""
    from unet_rir_amd import engine, data
    from oracle import torch_ref as R
    lo, hi = int(GOLD["global_min"]), int(GOLD["global_max"])
    assert engine.VOCAB == 2000 and R.VOCAB == 2000 and hi < engine.VOCAB and lo >= 0
    _, emb, _ = next(data.synthetic_batches(1, 64, 16, 16, "cpu"))
    assert tuple(emb.shape) == (64, 2, 16) and int(emb.min()) >= lo and int(emb.max()) <= hi
    _, emb_o, _ = R.synthetic_batch(R.Config(16, 16), 64)
    assert emb_o.min() >= lo and emb_o.max() <= hi
    import bench
    _, emb_b, _ = bench.synthetic_batch(64, 16, 16, torch.device("cpu"), 1234)
    assert int(emb_b.min()) >= lo and int(emb_b.max()) <= hi
    draws = torch.cat([next(data.synthetic_batches(1, 512, 8, 8, "cpu", seed=s))[1].reshape(-1) for s in range(4)])
    assert int(draws.min()) == lo and int(draws.max()) == hi          # both ends of the range are reachable
