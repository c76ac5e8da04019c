"""
Seeded synthetic feature generators.  Real covers80 /
DA-TACOS feature files are not available offline, so the benchmark and the
parity tests run on shape-matched synthetic chroma:

  * rand_set   -- the THROUGHPUT set: i.i.d. U[0,1) frames, each divided by its
                  max (look-alike of HPCP's per-frame max normalisation), fed
                  to similarity() as already-pooled features.
  * cover_set  -- the PARITY set: every work is a piecewise-constant chord
                  sequence; each version applies a circular bin shift, a tempo
                  warp and fresh noise, so that cliques are recoverable and MAP
                  is meaningful.
"""
import numpy as np

NBINS = 12


def _frame_max_normalise(x):
    mx = x.max(axis=1, keepdims=True)
    mx[mx == 0] = 1
    return (x / mx).astype(np.float32)


def pack(tracks):
    """list of (T_i, dim) arrays -> (frames (sum T, dim) f32, offsets (n+1) int64)."""
    lens = np.array([t.shape[0] for t in tracks], dtype=np.int64)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    frames = np.ascontiguousarray(np.concatenate(tracks, axis=0), dtype=np.float32)
    return frames, offsets


def rand_set(n_tracks, T=2000, seed=1234):
    rng = np.random.default_rng(seed)
    tracks = [_frame_max_normalise(rng.random((T, NBINS))) for _ in range(n_tracks)]
    frames, offsets = pack(tracks)
    return dict(frames=frames, offsets=offsets, labels=[str(i) for i in range(n_tracks)])


def _triads():
    t = np.zeros((24, NBINS))
    for root in range(12):
        for k, third in enumerate((4, 3)):
            v = np.zeros(NBINS)
            v[root] = 1.0
            v[(root + third) % 12] = 0.8
            v[(root + 7) % 12] = 0.9
            t[2 * root + k] = v
    return t


def cover_set(n_works=None, versions=5, seed=4321, t_range=(300, 600), clique_sizes=None, noise=0.05, segment_keep=1.0):
    """clique_sizes overrides (n_works, versions).  Returns frames/offsets/labels.
    noise: amplitude of the uniform noise added to every version; segment_keep < 1: every version keeps
    only a random contiguous part of the work (harder sets, so that MAP is not trivially 1)."""
    rng = np.random.default_rng(seed)
    tri = _triads()
    if clique_sizes is None:
        clique_sizes = [versions] * n_works
    tracks, labels = [], []
    for w, nv in enumerate(clique_sizes):
        T = int(rng.integers(t_range[0], t_range[1] + 1))
        chords = []
        c = int(rng.integers(0, 24))
        while len(chords) < T:
            seg = int(rng.integers(4, 17))
            chords += [c] * seg
            c = (c + int(rng.choice([-5, -2, 2, 5, 7, 1]))) % 24
        base = tri[np.array(chords[:T])]
        for _ in range(nv):
            shift = int(rng.integers(0, 12))
            fac = float(rng.uniform(0.8, 1.25))
            Tv = max(32, int(round(T * fac)))
            pos = np.linspace(0, T - 1, Tv)
            lo = np.floor(pos).astype(int)
            hi = np.minimum(lo + 1, T - 1)
            fr = (pos - lo)[:, None]
            x = (1 - fr) * base[lo] + fr * base[hi]
            x = np.roll(x, shift, axis=1) + noise * rng.random((Tv, NBINS))
            if segment_keep < 1.0:
                keep = max(32, int(round(Tv * segment_keep)))
                st = int(rng.integers(0, Tv - keep + 1))
                x = x[st:st + keep]
            tracks.append(_frame_max_normalise(x))
            labels.append("w%d" % w)
    frames, offsets = pack(tracks)
    return dict(frames=frames, offsets=offsets, labels=labels)


COVERS80_CLIQUES = [2] * 77 + [3] * 2 + [4]   # 164 tracks / 80 works (covers80_annotations.csv)


def covers80_shaped(seed=4321, t_range=(300, 600), noise=0.05, segment_keep=1.0):
    return cover_set(clique_sizes=COVERS80_CLIQUES, seed=seed, t_range=t_range, noise=noise, segment_keep=segment_keep)


def simple_raw_set(n_works, versions=3, seed=4321, t0_range=(15000, 25000)):
    """Raw (un-pooled) chroma for SiMPle: T0 frames so that WIN/SKIP pooling gives 150-250."""
    return cover_set(n_works=n_works, versions=versions, seed=seed, t_range=t0_range)


def earlyfusion_set(n_tracks, seed=4321, nb_range=(300, 500)):
    """Block features with the shapes of earlyfusion_traile.py:100-154."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_tracks):
        nb = int(rng.integers(nb_range[0], nb_range[1] + 1))
        mf = rng.standard_normal((nb, 650)).astype(np.float32)
        mf /= np.linalg.norm(mf, axis=1, keepdims=True)
        out.append(dict(mfccs=mf.astype(np.float32),
                        ssms=(2 * rng.random((nb, 1225))).astype(np.float32),
                        chromas=rng.random((nb, 480)).astype(np.float32),
                        chroma_med=rng.random(12)))
    return out


def earlyfusion_cover_set(n_works=30, versions=5, seed=4321, nb_range=(60, 100), noise=0.6, clique_sizes=None):
    """Cover-structured block features (the structure of cover_set carried into EarlyFusion's block space):
    a work is a piecewise-constant sequence of "states" (segments of 3-8 blocks), every state one random
    point per feature (mfcc block (650,), ssm (1225,), chroma block (40 x 12)); a version re-times the
    work (factor U[0.8, 1.25], nearest block), transposes the chroma (circular shift of the 12 bins of
    every frame and of chroma_med) and adds fresh noise of relative amplitude `noise`.  Cliques are
    recoverable but not trivially (MAP < 1 at the default noise).  Returns (tracks, labels)."""
    rng = np.random.default_rng(seed)
    if clique_sizes is None:
        clique_sizes = [versions] * n_works
    tracks, labels = [], []
    for w, nv in enumerate(clique_sizes):
        nb = int(rng.integers(nb_range[0], nb_range[1] + 1))
        nstates = int(rng.integers(6, 12))
        S = dict(mfccs=rng.standard_normal((nstates, 650)), ssms=2 * rng.random((nstates, 1225)),
                 chromas=rng.random((nstates, 40, 12)) ** 3)
        seq = []
        st = int(rng.integers(0, nstates))
        while len(seq) < nb:
            seq += [st] * int(rng.integers(3, 9))
            st = (st + int(rng.integers(1, nstates))) % nstates
        seq = np.array(seq[:nb])
        # slow drift inside a segment so that consecutive blocks of one state are close but not identical
        drift = np.cumsum(rng.standard_normal(nb)) * 0.05
        for _ in range(nv):
            fac = float(rng.uniform(0.8, 1.25))
            nbv = max(24, int(round(nb * fac)))
            pos = np.clip(np.round(np.linspace(0, nb - 1, nbv)).astype(int), 0, nb - 1)
            sv, dv = seq[pos], drift[pos]
            shift = int(rng.integers(0, 12))
            mf = S["mfccs"][sv] * (1.0 + dv[:, None]) + noise * rng.standard_normal((nbv, 650))
            mf /= np.linalg.norm(mf, axis=1, keepdims=True)
            ss = S["ssms"][sv] * (1.0 + dv[:, None]) + noise * 2 * rng.random((nbv, 1225))
            ch = S["chromas"][sv] + noise * 0.5 * rng.random((nbv, 40, 12))
            ch = np.roll(ch, shift, axis=2)
            tracks.append(dict(mfccs=mf.astype(np.float32), ssms=ss.astype(np.float32),
                               chromas=ch.reshape(nbv, 480).astype(np.float32),
                               chroma_med=np.median(ch.reshape(-1, 12), axis=0)))
            labels.append("w%d" % w)
    return tracks, labels
