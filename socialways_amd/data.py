"""Host-side data glue of the training loop (reference train.py:86-124, 446-461 and the dataset
schema of create_toy.py:187 / create_dataset.py:14): npz {obsvs (N,To,2), preds (N,Tp,2), times (N,),
batches (S,2)} -> normalised device tensors + scene ranges, the greedy scene packing of train(),
and synthetic generators for the benchmark shapes (no dataset ships with the reference).
"""
import numpy as np
import torch


class Scale(object):
    """Min-max, keep-ratio normalisation (utils/parse_utils.py:11-76)."""

    def __init__(self):
        self.min_x, self.max_x = +np.inf, -np.inf
        self.min_y, self.max_y = +np.inf, -np.inf
        self.sx, self.sy = 1, 1

    def calc_scale(self, keep_ratio=True):
        self.sx = 1 / (self.max_x - self.min_x)
        self.sy = 1 / (self.max_y - self.min_y)
        if keep_ratio:
            self.sx = self.sy = min(self.sx, self.sy)

    def normalize(self, data, shift=True, inPlace=True):
        out = data if inPlace else np.copy(data)
        out[..., 0] = (data[..., 0] - self.min_x * shift) * self.sx
        out[..., 1] = (data[..., 1] - self.min_y * shift) * self.sy
        return out

    def denormalize(self, data, shift=True, inPlace=False):
        out = data if inPlace else np.copy(data)
        out[..., 0] = data[..., 0] / self.sx + self.min_x * shift
        out[..., 1] = data[..., 1] / self.sy + self.min_y * shift
        return out


class SceneDataset:
    """train.py:89-124: 4/5 of the scenes train, the rest test; float32 keep-ratio normalisation;
    tracks resident on the device for the whole run.  `batches` may be int16 on disk
    (utils/parse_utils.py:490 overflows above 32767 rows): widened to int64 here."""

    def __init__(self, obsvs, preds, batches, times=None, device="cuda"):
        obsvs = np.array(obsvs, dtype=np.float32, copy=True)
        preds = np.array(preds, dtype=np.float32, copy=True)
        self.the_batches = np.asarray(batches).astype(np.int64)
        self.times = None if times is None else np.asarray(times)
        self.train_size = max(1, (len(self.the_batches) * 4) // 5)
        # train.py:95-98: both splits are taken BEFORE the single-scene substitution below, which only feeds the
        # look-ahead `the_batches[ii + 1]` of the packing loop (a one-scene dataset has NO test scene: test() = zeros)
        self.train_batches = self.the_batches[:self.train_size]
        self.test_batches = self.the_batches[self.train_size:]
        self.n_past, self.n_next = obsvs.shape[1], preds.shape[1]
        self.n_train_samples = int(self.the_batches[self.train_size - 1][1])
        self.n_test_samples = obsvs.shape[0] - self.n_train_samples
        if self.n_test_samples == 0:                                         # train.py:107-109
            self.n_test_samples = 1
            self.the_batches = np.array([self.the_batches[0], self.the_batches[0]])
        sc = Scale()
        sc.max_x = max(np.max(obsvs[:, :, 0]), np.max(preds[:, :, 0]))
        sc.min_x = min(np.min(obsvs[:, :, 0]), np.min(preds[:, :, 0]))
        sc.max_y = max(np.max(obsvs[:, :, 1]), np.max(preds[:, :, 1]))
        sc.min_y = min(np.min(obsvs[:, :, 1]), np.min(preds[:, :, 1]))
        sc.calc_scale(keep_ratio=True)
        self.scale, self.ss = sc, sc.sx
        self.obsv = torch.from_numpy(sc.normalize(obsvs)).to(device)
        self.pred = torch.from_numpy(sc.normalize(preds)).to(device)

    @classmethod
    def from_npz(cls, path, device="cuda"):
        d = np.load(path)
        return cls(d["obsvs"], d["preds"], d["batches"], d["times"] if "times" in d.files else None, device)

    def packed_steps(self, batch_size):
        """Greedy scene packing of train() (train.py:446-456): scenes are appended until the next one
        would exceed `batch_size` AGENTS.  Yields (row_start, row_end, sub_batches)
        with sub_batches already relative to row_start (train.py:461)."""
        tb, allb = self.train_batches, self.the_batches
        acc, subs = 0, []
        for ii, b in enumerate(tb):
            acc += int(b[1] - b[0])
            subs.append(b)
            if ii >= self.train_size - 1 or acc + int(allb[ii + 1][1] - allb[ii + 1][0]) > batch_size:
                a = int(subs[0][0])
                yield a, int(subs[-1][1]), np.asarray(subs, dtype=np.int64) - a
                acc, subs = 0, []


def synth_tracks(n_scenes, agents, n_past=8, n_next=12, seed=1234):
    """Synthetic crowd of SURVEY §8d: p0~U[0,10)^2, v = N(0,0.3^2) (per agent) + cumsum_t N(0,0.05^2),
    track = p0 + cumsum_t v, float32, scenes contiguous.  `agents`: int or per-scene list."""
    rng = np.random.default_rng(seed)
    sizes = [int(agents)] * n_scenes if np.isscalar(agents) else [int(a) for a in agents]
    N, T = int(np.sum(sizes)), n_past + n_next
    p0 = rng.uniform(0, 10, size=(N, 1, 2))
    v = rng.normal(0, 0.3, size=(N, 1, 2)) + np.cumsum(rng.normal(0, 0.05, size=(N, T, 2)), axis=1)
    track = (p0 + np.cumsum(v, axis=1)).astype(np.float32)
    ends = np.cumsum(sizes)
    batches = np.stack([ends - np.asarray(sizes), ends], axis=1).astype(np.int64)
    times = np.repeat(np.arange(len(sizes)), sizes).astype(np.int32)
    return dict(obsvs=track[:, :n_past], preds=track[:, n_past:], times=times, batches=batches)


def ragged_scene_sizes(n_agents=2048, max_agents=8, seed=77):
    """Scene sizes of a real-shaped packed batch: ETH/UCY recordings hold 1..8 pedestrians per timestamp and
    train.py:446-461 packs whole scenes up to --batch-size agents; uniform 1..max_agents until the batch holds exactly
    `n_agents` agents (single-agent scenes included)."""
    rng = np.random.default_rng(seed)
    sizes = []
    while sum(sizes) < n_agents:
        sizes.append(int(rng.integers(1, max_agents + 1)))
    sizes[-1] -= sum(sizes) - n_agents
    return [a for a in sizes if a > 0]


def toy_tracks(n_samples=768, n_conditions=8, n_modes=3, n_per_batch=6, seed=30):
    """The toy multi-modal set of create_toy.py:11-54 + its npz packing (:162-187): 4-point tracks
    (2 observed + 2 to predict) approaching the origin from `n_conditions` directions and turning
    by one of `n_modes` angles.  Draw order per sample: one uniform for the 3rd point, one for the
    4th, numpy legacy seed 30 (create_toy.py:37-48,145)."""
    rs = np.random.RandomState(seed)
    samples, stamps = np.zeros((n_samples, 4, 2)), np.zeros(n_samples)
    for ii in range(n_samples):
        way = (ii * n_conditions) // n_samples
        w_i = way % (n_conditions / n_per_batch)
        t0 = ii % (n_samples // n_conditions) + w_i * (n_samples // n_conditions)
        ang = way * (2.0 * np.pi / n_conditions)
        turn = ((ii % n_modes) - n_modes // 2) * 16 * np.pi / 180
        r2 = (rs.rand() - 0.5) * 4 * np.pi / 180
        r3 = (rs.rand() - 0.5) * 6 * np.pi / 180
        samples[ii] = [[np.cos(ang) * 4, np.sin(ang) * 4], [np.cos(ang) * 3, np.sin(ang) * 3],
                       [np.cos(ang + turn + r2) * 2, np.sin(ang + turn + r2) * 2],
                       [np.cos(ang + turn + r2 + r3), np.sin(ang + turn + r2 + r3)]]
        stamps[ii] = t0 * 4
    samples = samples / 4
    groups = {}
    for ii in range(n_samples):                                              # scene = identical first timestamp
        groups.setdefault(stamps[ii], []).append(ii)
    obsvs, preds, times, batches = [], [], [], []
    for key, idx in groups.items():
        batches.append([len(obsvs), len(obsvs) + len(idx)])
        for k in idx:
            obsvs.append(samples[k][:2])
            preds.append(samples[k][2:])
            times.append(stamps[k])
    return dict(obsvs=np.array(obsvs).astype(np.float32), preds=np.array(preds).astype(np.float32),
                times=np.array(times).astype(np.int32), batches=np.array(batches))


def shard_scenes(sub_batches, world_size):
    """Contiguous, scene-aligned split of one packed batch over `world_size` ranks (never splits a
    scene), balanced by a per-scene cost of agents + pairs/8 (the social block is O(n^2), the LSTM
    / decoder O(n)).  Returns [(scene_lo, scene_hi)] per rank; trailing ranks may get (k,k)."""
    sb = np.asarray(sub_batches, dtype=np.int64).reshape(-1, 2)
    n = sb[:, 1] - sb[:, 0]
    cost = n + (n * n) / 8.0
    csum = np.concatenate([[0.0], np.cumsum(cost)])
    total = csum[-1]
    cuts = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        k = int(np.searchsorted(csum, target, side="left"))
        if k > 0 and abs(csum[k - 1] - target) <= abs(csum[min(k, len(csum) - 1)] - target):
            k -= 1
        cuts.append(min(max(k, cuts[-1]), len(sb)))
    cuts.append(len(sb))
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


# ------------------------------------------------------------------------------------------------
# Dataset preparation: BIWI/ETH "obsmat.txt" -> sliding windows -> scenes.
# Reference: utils/parse_utils.py:231-320 (BIWIParser.load), :457-508 (create_dataset),
# create_dataset.py:1-15.  Offline, host-side, numpy.
# ------------------------------------------------------------------------------------------------
def write_biwi_obsmat(path, frames, ids, pos, vel=None):
    """Write tracks in the BIWI column layout [frame, id, px, pz, py, vx, vz, vy] (the parser reads
    columns 0,1,2,4,5,7; parse_utils.py:273-283).  Rows are written in the order given."""
    frames, ids, pos = np.asarray(frames), np.asarray(ids), np.asarray(pos, dtype=np.float64)
    vel = np.zeros_like(pos) if vel is None else np.asarray(vel, dtype=np.float64)
    with open(path, "w") as f:
        for t, i, p, v in zip(frames, ids, pos, vel):
            f.write("%.7e %.7e %.7e %.7e %.7e %.7e %.7e %.7e\n" % (t, i, p[0], 0.0, p[1], v[0], 0.0, v[1]))


def parse_biwi(path, down_sample=1):
    """BIWIParser.load for a single file: per-pedestrian position / time arrays in first-appearance
    order and the frame interval of the first pedestrian with two samples (parse_utils.py:300-305).
    Returns (p_data, t_data, interval)."""
    delim = "\t" if "zara" in path else " "
    order, pos, tim = [], {}, {}
    with open(path) as f:
        for line in f:
            row = [c for c in line.split(delim) if c != ""]
            if len(row) < 8:
                continue
            ts = float(row[0])
            pid = round(float(row[1]))
            if ts % down_sample != 0:
                continue
            if pid not in pos:
                order.append(pid)
                pos[pid], tim[pid] = [], []
            pos[pid].append((float(row[2]), float(row[4])))
            tim[pid].append(ts)
    interval = -1
    for pid in order:
        if len(tim[pid]) > 1:
            d = int(round(tim[pid][1] - tim[pid][0]))
            if d > 0:
                interval = d
                break
    p_data = [np.array(pos[k]) for k in order]
    t_data = [np.array(tim[k]).astype(np.int32) for k in order]
    return p_data, t_data, interval


def create_dataset(p_data, t_data, t_range, n_past=8, n_next=12):
    """Sliding windows of n_past + n_next samples (parse_utils.py:457-508): for every integer t of
    `t_range` (stepping by 1, like the reference) and every pedestrian that has samples at
    t - step*n_past, t and t + step*(n_next-1): obs = the n_past samples before t, pred = the n_next
    from t on.  A scene = the windows sharing t.  Returns (obsvs f32, preds f32, times list,
    batches int64); the reference stores `batches` as int16 (overflow above 32767 rows)."""
    step = t_range.step
    index = [dict((int(t), k) for k, t in reversed(list(enumerate(td)))) for td in t_data]   # first occurrence wins
    t0s, xs, ys = [], [], []
    for t in range(t_range.start, t_range.stop, 1):
        for i, ix in enumerate(index):
            a, b, c = ix.get(t - step * n_past), ix.get(t), ix.get(t + step * (n_next - 1))
            if a is None or b is None or c is None:
                continue
            t0s.append(t)
            xs.append(p_data[i][a:b])
            ys.append(p_data[i][b:c + 1])
    batches, keep, last_t = [], [], -1000
    for i, t in enumerate(t0s):          # min_interval = 1 (parse_utils.py:481-489)
        if t > last_t + 1:
            batches.append([i, i + 1])
            last_t = t
        if t == last_t:
            batches[-1][1] = i + 1
    out_b, last = [], 0
    for a, b in batches:
        keep += list(range(a, b))
        out_b.append([last, last + (b - a)])
        last += b - a
    obsvs = np.array([xs[k] for k in keep]).astype(np.float32)
    preds = np.array([ys[k] for k in keep]).astype(np.float32)
    return obsvs, preds, t0s, np.array(out_b, dtype=np.int64)


def biwi_to_npz(obsmat_path, npz_path, n_past=8, n_next=12):
    """create_dataset.py:1-15: obsmat.txt -> npz {obsvs, preds, times, batches}."""
    p_data, t_data, interval = parse_biwi(obsmat_path)
    obsvs, preds, times, batches = create_dataset(p_data, t_data, range(int(t_data[0][0]), int(t_data[-1][-1]), interval),
                                                  n_past, n_next)
    np.savez(npz_path, obsvs=obsvs, preds=preds, times=times, batches=batches)
    return obsvs, preds, times, batches


def synth_crowd_frames(n_frames=60, n_ped=24, interval=6, seed=3, max_life=40):
    """A synthetic ETH-like recording (no ETH/UCY data is available): pedestrians enter at
    random frames, walk with slowly varying velocity for a random life span; returned frame-major
    like obsmat.txt (frames, ids, positions, velocities)."""
    rng = np.random.default_rng(seed)
    rows = []
    for pid in range(1, n_ped + 1):
        t_in = int(rng.integers(0, max(1, n_frames - 22)))
        life = int(rng.integers(21, max_life))
        p = rng.uniform(-5, 5, size=2)
        v = rng.normal(0, 0.4, size=2)
        for k in range(min(life, n_frames - t_in)):
            rows.append(((t_in + k) * interval, pid, p.copy(), v.copy()))
            v = v + rng.normal(0, 0.03, size=2)
            p = p + v * 0.4
    rows.sort(key=lambda r: (r[0], r[1]))
    return (np.array([r[0] for r in rows], dtype=np.float64), np.array([r[1] for r in rows], dtype=np.float64),
            np.array([r[2] for r in rows]), np.array([r[3] for r in rows]))
