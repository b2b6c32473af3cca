"""Generates tests/golden/rooms_emb.npz by IMPORTING the reference's rooms.py (pure `math`; the only reference file that is
importable in the build container - everything else needs TensorFlow) and evaluating UTSRoom.return_embedding
(rooms.py:96-99) for the six rooms dataset.py:86-91 defines.  Run in the build container only (the reference does not
travel to the GPU box); the .npz it writes is data: inputs (room parameters, characteristics) and expected outputs.

It is the one reference-held pin this path has: the value contract of the `emb` input (16 integers per position, the
range Embedding(2000, 256) must cover - dl_models/u_net.py:257), as DataGenerator.__getitem__ stacks it (datageneratorv2.py:89).
"""
import importlib.util
import os
import sys

import numpy as np

REF = "/root/reference/rooms.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rooms_emb.npz")

# dataset.py:86-91 (constructor arguments of the six rooms; data, not code)
ROOMS = {
    "AnechoicRoom": (490, 722, 490, 722, 90, 90, 90, 90, 529, [245, 361], 45),
    "HemiAnechoicRoom": (490, 722, 490, 722, 90, 90, 90, 90, 529, [245, 361], 52),
    "SmallMeetingRoom": (355, 410, 401, 378, 96, 90, 85, 88, 300, [175.5, 205], 497),
    "MediumMeetingRoom": (736, 520, 650, 434.5, 81, 92, 98, 89, 300, [368, 217.5], 659),
    "LargeMeetingRoom": (994, 923, 1087, 1022, 81.4, 105, 81.3, 92.3, 300, [497, 486.25], 1281),
    "ShoeBoxRoom": (600, 1175, 600, 1175, 90, 90, 90, 90, 300, [300, 881.25], 667),
}
ZONES = ["A", "B", "C", "D", "E"]
ARRAYS = {"Planar": 64, "Circular": 60}


def main():
    spec = importlib.util.spec_from_file_location("ref_rooms", REF)
    rooms = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rooms)
    names = sorted(ROOMS)
    objs = {n: rooms.UTSRoom(*ROOMS[n]) for n in names}
    lo, hi, count = 10 ** 9, -10 ** 9, 0
    per_room_min, per_room_max = [], []
    for n in names:
        rlo, rhi = 10 ** 9, -10 ** 9
        for z in ZONES:
            for a, n_mic in ARRAYS.items():
                for l in range(1, 61):
                    for m in range(1, n_mic + 1):
                        v = objs[n].return_embedding([n, z, a, str(l), str(m)])
                        assert len(v) == 16
                        rlo, rhi = min(rlo, min(v)), max(rhi, max(v))
                        count += 1
        per_room_min.append(rlo); per_room_max.append(rhi)
        lo, hi = min(lo, rlo), max(hi, rhi)
    # a handful of (room, zone, array, speaker, microphone) -> 16 ints samples, incl. the one SURVEY.md quotes
    samples = [("LargeMeetingRoom", "B", "Circular", 22, 1), ("LargeMeetingRoom", "A", "Planar", 1, 1),
               ("SmallMeetingRoom", "E", "Planar", 60, 64), ("MediumMeetingRoom", "C", "Circular", 30, 45),
               ("ShoeBoxRoom", "D", "Planar", 7, 33), ("HemiAnechoicRoom", "A", "Circular", 59, 60),
               ("AnechoicRoom", "E", "Circular", 15, 31), ("SmallMeetingRoom", "B", "Circular", 1, 30)]
    vecs = np.array([objs[r].return_embedding([r, z, a, str(l), str(m)]) for r, z, a, l, m in samples], dtype=np.int64)
    np.savez(OUT, sample_room=np.array([s[0] for s in samples]), sample_zone=np.array([s[1] for s in samples]),
             sample_array=np.array([s[2] for s in samples]), sample_speaker=np.array([s[3] for s in samples]),
             sample_mic=np.array([s[4] for s in samples]), sample_emb=vecs,
             room_names=np.array(names), room_args=np.array([list(ROOMS[n][:9]) + ROOMS[n][9] + [ROOMS[n][10]] for n in names], dtype=np.float64),
             room_min=np.array(per_room_min), room_max=np.array(per_room_max), global_min=lo, global_max=hi, n_combinations=count)
    print(f"{count} combinations, range [{lo}, {hi}] -> {OUT}")


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("the reference is not present here (build container only)")
    main()
