"""The `emb` input of the model: 16 integers per (room, zone, array, loudspeaker, microphone) position, as
`Dataset.get_embedding` -> `UTSRoom.return_embedding` build them (dataset.py:185-212, rooms.py:46-99): nine room numbers
(rounded side lengths a-d, corner angles, height) followed by the loudspeaker position, the microphone position (cm, grid
coordinates) and the room's RT60 in ms.  `DataGenerator.__getitem__` stacks the vector of the input position and of the
target position into the int32 `[B, 2, 16]` tensor the information-vector branch embeds (datageneratorv2.py:89;
dl_models/u_net.py:253-263, Embedding(2000, 256)).

Table-driven restatement of that geometry for callers that assemble batches without the reference's dataset class; pinned
value for value by tests/golden/rooms_emb.npz, which is generated from the reference's own rooms.py.
"""
import math

import numpy as np

# (a, b, c, d, alpha, beta, gamma, delta, height, grid centre (x, y), rt60) - dataset.py:86-91
UTS_ROOMS = {
    "AnechoicRoom": (490, 722, 490, 722, 90, 90, 90, 90, 529, (245, 361), 45),
    "HemiAnechoicRoom": (490, 722, 490, 722, 90, 90, 90, 90, 529, (245, 361), 52),
    "SmallMeetingRoom": (355, 410, 401, 378, 96, 90, 85, 88, 300, (175.5, 205), 497),
    "MediumMeetingRoom": (736, 520, 650, 434.5, 81, 92, 98, 89, 300, (368, 217.5), 659),
    "LargeMeetingRoom": (994, 923, 1087, 1022, 81.4, 105, 81.3, 92.3, 300, (497, 486.25), 1281),
    "ShoeBoxRoom": (600, 1175, 600, 1175, 90, 90, 90, 90, 300, (300, 881.25), 667),
}
ZONE_OFFSET = {"A": (-40, 0), "B": (40, 0), "C": (0, 40), "D": (0, -40), "E": (0, 0)}      # cm from the grid centre
EAR_HEIGHT = 145          # z of loudspeakers and microphones
SPEAKER_RADIUS = 150      # 60 loudspeakers on a circle, 6 degrees apart, the first at +3 degrees
EMB_MIN, EMB_MAX = 26, 1281          # value range over every position of the set (fixture: global_min / global_max)


def uts_room_embedding(room, zone, array, speaker, mic):
    """-> list of 16 ints for one position.  `array` is 'Planar' (8 x 8 grid, 4 cm pitch, microphones 1..64 row by row) or
    'Circular' (two rings of 30: radius 12 cm for microphones 1..30, 10 cm for 31..60)."""
    a, b, c, d, al, be, ga, de, height, (cx, cy), rt60 = UTS_ROOMS[room]
    l, m = int(speaker), int(mic)
    ang = (2 * l - 1) * math.pi / 60
    xl = round(-SPEAKER_RADIUS * math.sin(ang)) + cx
    yl = round(SPEAKER_RADIUS * math.cos(ang)) + cy
    ox, oy = ZONE_OFFSET[zone]
    if array == "Planar":
        col, row = (m - 1) % 8, (m - 1) // 8
        xm, ym = -14 + 4 * col + ox + cx, 14 - 4 * row + oy + cy
    elif array == "Circular":
        radius = 12 - 2 * ((m - 1) // 30)
        phi = ((m - 1) % 30) * 2 * math.pi / 30
        xm, ym = -radius * math.sin(phi) + ox + cx, radius * math.cos(phi) + oy + cy
    else:
        raise ValueError("array must be 'Planar' or 'Circular'")
    return [round(a), round(b), round(c), round(d), round(al), round(be), round(ga), round(de), round(height),
            round(xl), round(yl), EAR_HEIGHT, round(xm), round(ym), EAR_HEIGHT, rt60]


def emb_pair(pos_in, pos_out):
    """int32 [2, 16]: the information vector of one training pair (input position, target position)."""
    return np.array([uts_room_embedding(*pos_in), uts_room_embedding(*pos_out)], dtype=np.int32)
