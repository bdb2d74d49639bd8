"""CPU tests of the library's host logic through host-only handles (device -1): topology (A1, bit-exact against
the oracle and numpy), tiles and packed words, degrees of freedom and thermostat masses (A2), error behaviour.
Nothing is launched (no GPU here)."""
import numpy as np
import pytest

from openmm_drudenose_amd import synth, _lib, DrudeTGNHIntegrator, HostTopology, TgnhError
from helpers import make_oracle, to_internal


def integ(chains=3, drude_chains=True, com=True, group=None, ngroups=0):
    it = DrudeTGNHIntegrator(300.0, 0.1, 1.0, 0.005, 0.001, 20, chains, drude_chains, com)
    for _ in range(ngroups):
        it.addTempGroup()
    if group is not None:
        for g in group:
            it.addParticleTempGroup(int(g))
    return it


BUILDERS = {"pnm": synth.pair_normal_massless, "nacl": synth.nacl, "il": lambda: synth.ionic_liquid(30),
            "mixed": lambda: synth.mixed(250, 12), "water": lambda: synth.water_box(700)}


@pytest.mark.parametrize("name", list(BUILDERS))
def test_topology_bit_exact_against_oracle(name):
    s, g, ng = BUILDERS[name]()
    it = integ(group=g, ngroups=ng)
    t = HostTopology(s, it, mode="TGNH")
    o = make_oracle(s, g, ng, "TGNH", it)
    assert np.array_equal(t.topology(0), o.normal_particles())                  # Ref :113-137
    assert np.array_equal(t.topology(1), s.pair_drude) and np.array_equal(t.topology(2), s.pair_parent)
    assert np.array_equal(t.topology(3), g) and np.array_equal(t.topology(4), s.resid)
    assert np.array_equal(t.topology(5), np.bincount(s.resid, minlength=s.num_residues))      # Cu :121
    first = np.full(s.num_residues, -1)
    for i in range(s.num_particles - 1, -1, -1):
        first[s.resid[i]] = i
    assert np.array_equal(t.topology(6), first)                                  # Cu :122-125
    ts = t.topology(7)
    assert ts[0] == 0 and ts[-1] == s.num_particles and np.all(np.diff(ts) > 0) and np.all(np.diff(ts) <= 512)
    tile_of = np.searchsorted(ts, np.arange(s.num_particles), side="right") - 1
    assert np.all(tile_of[s.pair_drude] == tile_of[s.pair_parent])
    assert np.all(tile_of[np.r_[0, np.flatnonzero(np.diff(s.resid)) + 1]] == tile_of[np.r_[np.flatnonzero(np.diff(s.resid)), s.num_particles - 1]])
    meta = t.topology(8).view(np.uint32)
    assert np.array_equal(np.flatnonzero((meta & 3) == 1), np.sort(s.pair_drude))
    assert np.array_equal(np.flatnonzero((meta & 3) == 2), np.sort(s.pair_parent))
    assert np.array_equal(((meta >> 2) & 255).astype(np.int32), g)
    off = ((meta >> 10) & 2047).astype(np.int64) - 1024
    assert np.array_equal((np.arange(s.num_particles) + off)[s.pair_drude], s.pair_parent)
    assert np.array_equal((np.arange(s.num_particles) + off)[s.pair_parent], s.pair_drude)
    # tile-local residue index
    res_first_of_tile = s.resid[ts[:-1]]
    assert np.array_equal((meta >> 21).astype(np.int64), s.resid - res_first_of_tile[tile_of])


@pytest.mark.parametrize("name,com", [("pnm", True), ("nacl", True), ("il", True), ("mixed", True), ("water", True), ("mixed", False),
                                      ("polymer", True), ("polymer", False)])
def test_wave_tiles_and_their_words(name, com):
    """The wave tiles of the kinetic-energy passes (wke_kernel / wstep_kernel): <= 64 consecutive slots, cut never through a
    Drude pair and -- when the molecular COM is needed -- never through a molecule; each entry carries the tile's largest
    molecule; the per-slot word decodes to role, group, partner, position in the molecule and its size.  A topology with a
    molecule longer than a wavefront (and a COM to form) has no wave tiles: its KE passes stay on the tile kernel."""
    s, g, ng = synth.polymer_in_water(150, 40) if name == "polymer" else BUILDERS[name]()       # (polymer: one 750-slot molecule)
    it = integ(group=g, ngroups=ng, com=com)
    t = HostTopology(s, it, mode="TGNH")
    wt = t.topology(9).reshape(-1, 2)
    n = s.num_particles
    longest = np.bincount(s.resid).max()
    if com and longest > 64:
        assert len(wt) == 0                                  # no wave tiles: the tile kernel keeps the KE passes
        return
    assert len(wt) >= 2 and wt[0, 0] == 0 and wt[-1, 0] == n
    start = wt[:, 0]
    assert np.all(np.diff(start) > 0) and np.all(np.diff(start) <= 64)
    tile_of = np.searchsorted(start, np.arange(n), side="right") - 1
    assert np.all(tile_of[s.pair_drude] == tile_of[s.pair_parent])
    w = t.topology(10).view(np.uint32)
    assert np.array_equal(np.flatnonzero((w & 3) == 1), np.sort(s.pair_drude))
    assert np.array_equal(np.flatnonzero((w & 3) == 2), np.sort(s.pair_parent))
    assert np.array_equal(((w >> 2) & 255).astype(np.int32), g)
    off = ((w >> 10) & 127).astype(np.int64) - 64
    assert np.array_equal((np.arange(n) + off)[s.pair_drude], s.pair_parent)
    assert np.array_equal((np.arange(n) + off)[s.pair_parent], s.pair_drude)
    assert np.all(off[(w & 3) == 0] == 0)
    pos, size = ((w >> 17) & 63).astype(np.int64), ((w >> 23) & 63).astype(np.int64) + 1
    if com:
        first = np.r_[0, np.flatnonzero(np.diff(s.resid)) + 1]
        count = np.diff(np.r_[first, n])
        mol = np.cumsum(np.r_[0, np.diff(s.resid) != 0])
        assert np.array_equal(pos, np.arange(n) - first[mol]) and np.array_equal(size, count[mol])
        assert np.all(tile_of[first] == tile_of[first + count - 1])             # no molecule is cut
        for k in range(len(wt) - 1):
            assert wt[k, 1] == size[start[k]:start[k + 1]].max()
    else:
        assert np.all(pos == 0) and np.all(size == 1) and np.all(wt[:-1, 1] == 1)


def _wave_tile_system(name):
    """the systems of test_every_access_of_the_wave_tile_kernels_stays_inside_its_buffers"""
    if name == "testwater":                                  # the reference's testWater box: 216 x (O, D, H1, H2, M), constraints + M site
        import water_test_system as wts
        s = wts.build()
        return s, np.zeros(s.num_particles, np.int32), 1
    if name.startswith("ragged"):
        from helpers import random_topology
        mass, pd, pp, resid, group, ngroups, cons, sizes, first, rng = random_topology(int(name[6:]))
        s = synth.DrudeSystem(mass=mass, pair_drude=np.array(pd, np.int32), pair_parent=np.array(pp, np.int32), resid=resid,
                              constraints=np.array(cons, np.int32).reshape(-1, 2))
        return s, group, ngroups
    return BUILDERS[name]()


@pytest.mark.parametrize("mode,com", [("TGNH", True), ("TGNH", False), ("dualNH", True)])
@pytest.mark.parametrize("name", ["testwater", "water", "nacl", "il", "mixed", "pnm"] + [f"ragged{k}" for k in range(8)])
def test_every_access_of_the_wave_tile_kernels_stays_inside_its_buffers(name, mode, com):
    """What wke_kernel / wstep_kernel touch, re-derived on the CPU from the tables a handle hands its launches
    (wave_load, load_vf, work / prepare / finish in tgnh_kernels.hip; launch sizes in tgnh_host.cpp):

      global   slot ws + lane for lane < n of every wave tile: inside [0, N); the table has num_wtiles + 1 entries, the last
               one N; velocities, index words and forces are read there and nowhere else;
      LDS      a lane reads its molecule's lanes first .. first + (molecule slots - 1) with first = lane - position, and its
               Drude partner's lane: all inside [0, n) of the wavefront's own 64-lane image; the walk's trip count (the tile's
               largest molecule) covers every molecule of the tile; a padding lane reads itself;
      rows     one row of partial sums per work-group in a table of GRID_CAP rows, or tagged cells (row_word's layout) in an
               area sized for GRID_CAP rows of <= 10 thermostats: the largest grid any launch of the handle takes fits both.

    On the reference's own testWater box (the box of the one device fault this repository has on record, HISTORY.md section 8b), the
    ragged topologies of helpers.random_topology, and the synthetic boxes; TGNH with and without the COM group, dualNH."""
    s, g, ng = _wave_tile_system(name)
    if mode == "dualNH":
        if s.num_pairs == 0:
            pytest.skip("dualNH needs a Drude pair")
        it = integ(com=com)
    else:
        it = integ(group=g, ngroups=ng, com=com)
    t = HostTopology(s, it, mode=mode)
    n = s.num_particles
    wt = t.topology(9).reshape(-1, 2)
    bounds = np.zeros(8, np.int32)
    assert t.lib.tgnh_get_launch_bounds(t.h, bounds.ctypes.data_as(_lib.c_i32p)) == 0
    tiles, wtiles, entries, grid, rows, tagged, touched, NT = (int(x) for x in bounds)
    uses_com = mode == "TGNH" and com
    longest = np.bincount(s.resid).max()
    # ---- launch-side sizes
    assert entries == len(wt) and (wtiles == 0) == (len(wt) == 0) and (wtiles == 0 or entries == wtiles + 1)
    assert 1 <= grid <= rows and rows == 2048
    assert grid >= min(tiles, 2048) and (wtiles == 0 or grid >= min((wtiles + 3) // 4, 2048))
    if NT <= 10:
        assert tagged == 2 * 2048 * 10 and 0 < touched <= tagged
        # row_word(r, j) = ((r >> 6) * 20 + j) * 64 + (r & 63): the last word of the last row of that grid
        r, j = grid - 1, 2 * NT - 1
        assert touched == ((r >> 6) * 20 + j) * 64 + (r & 63) + 1
    else:
        assert tagged == 0 and touched == 0                 # more than 8 groups: no tagged rows, the tile kernels' partial rows only
    # cuts that would go through a pair or (COM group on) a molecule; no legal cut within 64 slots somewhere = no wave tiles
    cut_ok = np.ones(n + 1, bool)
    for a, b in zip(s.pair_drude, s.pair_parent):
        cut_ok[min(a, b) + 1:max(a, b) + 1] = False
    if uses_com:
        cut_ok[1:n] &= np.diff(s.resid) != 0
    cut_ok[n] = True
    pos, possible = 0, not (uses_com and longest > 64)
    while possible and pos < n:
        ends = np.flatnonzero(cut_ok[pos + 1:min(pos + 64, n) + 1])
        possible = len(ends) > 0
        pos = pos + 1 + int(ends[-1]) if possible else pos
    assert (wtiles > 0) == possible, (name, mode, com)
    if not possible:
        return
    # ---- the table
    start = wt[:, 0].astype(np.int64)
    assert start[0] == 0 and start[-1] == n and np.all(np.diff(start) >= 1) and np.all(np.diff(start) <= 64)
    w = t.topology(10).view(np.uint32)
    assert len(w) == n
    for k in range(wtiles):
        ws, cnt, maxn = int(start[k]), int(start[k + 1] - start[k]), int(wt[k, 1])
        lane = np.arange(64)
        word = np.where(lane < cnt, np.r_[w[ws:ws + cnt], np.zeros(64 - cnt, np.uint32)], np.uint32(64 << 10))   # wave_load's padding word
        assert ws + cnt <= n                                                  # every global index of the tile
        j, n1 = ((word >> 17) & 63).astype(np.int64), ((word >> 23) & 63).astype(np.int64)
        first = lane - j
        live = lane < cnt
        if uses_com:
            assert np.all(first[live] >= 0) and np.all((first + n1)[live] < cnt), (name, k)     # the molecule lies inside the tile
            assert np.all(n1[live] + 1 <= maxn), (name, k)                   # the walk (k < maxn, k <= n1) reaches every slot of it
        assert np.all(j[~live] == 0) and np.all(n1[~live] == 0)               # padding lanes: a molecule of their own
        pl = lane + ((word >> 10) & 127).astype(np.int64) - 64
        role = word & 3
        assert np.all((pl >= 0) & (pl < 64)) and np.all(pl[live & (role != 0)] < cnt), (name, k)
        assert np.all(pl[role == 0] == lane[role == 0])
        # partners point at each other, Drude <-> parent
        pr = role[pl]
        assert np.all(pr[role == 1] == 2) and np.all(pr[role == 2] == 1) and np.all(pl[pl[role != 0]] == lane[role != 0])
    t.close()


@pytest.mark.parametrize("name,com", [("pnm", True), ("nacl", True), ("il", True), ("mixed", True), ("water", True), ("mixed", False),
                                      ("water", False), ("polymer", True), ("polymer", False)])
def test_tiles_of_identical_molecules_need_no_per_slot_words(name, com):
    """A tile that repeats one kind of molecule (or a few) is marked at create (topology 11 / 12) and its kernels form the
    per-slot word from the slot's position -- pattern[k mod P], plus the molecule's index (k div P) * molecules per period << 21
    in the 512-slot tiles -- instead of reading 4 B per slot.  Every marked tile's pattern reproduces its words exactly; a water box is marked
    throughout, the mixed boxes where their waters are."""
    s, g, ng = synth.polymer_in_water(150, 40) if name == "polymer" else BUILDERS[name]()
    it = integ(group=g, ngroups=ng, com=com)
    t = HostTopology(s, it, mode="TGNH")
    n = s.num_particles
    meta, starts = t.topology(8).view(np.uint32), t.topology(7)
    pat, words = t.topology(11).view(np.uint32), t.topology(13).view(np.uint32).reshape(-1, 64)
    assert len(pat) == max(len(starts) - 1, 1)
    marked = 0
    for k in range(len(starts) - 1):
        ts, m = starts[k], starts[k + 1] - starts[k]
        P, mols, pid = int(pat[k] & 255), int((pat[k] >> 8) & 255), int(pat[k] >> 16)
        if P == 0:
            continue
        marked += m
        pos = np.arange(m)
        made = words[pid][pos % P] + (((pos // P) * mols).astype(np.uint32) << 21 if com else 0)
        assert np.array_equal(made.astype(np.uint32), meta[ts:ts + m]), (name, com, k)
    wt = t.topology(9).reshape(-1, 2)
    wmarked = 0
    if len(wt):
        wmeta = t.topology(10).view(np.uint32)
        wpat, wwords = t.topology(12).view(np.uint32), t.topology(14).view(np.uint32).reshape(-1, 64)
        assert len(wpat) == len(wt) - 1
        for k in range(len(wt) - 1):
            ws, m = wt[k, 0], wt[k + 1, 0] - wt[k, 0]
            P, pid = int(wpat[k] & 255), int(wpat[k] >> 8)
            if P == 0:
                continue
            wmarked += m
            assert np.array_equal(wwords[pid][np.arange(m) % P], wmeta[ws:ws + m]), (name, com, k)
    print(name, com, "slots in marked tiles:", marked, "of", n, "; in marked wave tiles:", wmarked)
    if name == "water":
        assert marked == n and wmarked == n
        if com:
            assert len(words) == 1 and len(wwords) == 1      # one kind of molecule, tiles cut between molecules: one pattern
    if name == "mixed":                                      # (every tenth water in a group of its own: periods of ten molecules)
        assert 0 < marked < n and 0 < wmarked < n


@pytest.mark.parametrize("mode", ["dualNH", "TGNH"])
@pytest.mark.parametrize("com,chains,drude_chains,cmm", [(True, 1, True, False), (True, 3, False, True), (False, 4, True, True),
                                                         (True, 6, True, False)])
def test_dof_and_thermostat_block_match_oracle(mode, com, chains, drude_chains, cmm):
    s, g, ng = synth.mixed(120, 9)
    s.has_cm_motion_remover = cmm
    if mode == "dualNH":
        g, ng = np.zeros_like(g), 1
    it = integ(chains=chains, drude_chains=drude_chains, com=com, group=g if mode == "TGNH" else None,
               ngroups=ng if mode == "TGNH" else 0)
    t = HostTopology(s, it, mode=mode)
    o = make_oracle(s, g, ng, mode, it)
    dof_o, nkt_o = o.dof()
    dof, nkt = t.dof()
    assert np.allclose(dof, to_internal(dof_o, mode), rtol=1e-14, atol=0)
    assert np.allclose(nkt, to_internal(nkt_o, mode), rtol=1e-14, atol=0)
    for which in range(4):                                      # eta, etaDot, etaDotDot, etaMass: same layout, same values
        a, b = t.thermostat_state(which), o.chain(which)
        assert a.shape == b.shape and np.allclose(a, b, rtol=1e-14, atol=0), which


def test_constraints_reduce_group_dof_and_must_not_span_groups():
    s, g, ng = synth.mixed(40, 3)
    # constrain O-H1, O-H2 of every water (slots 0,2 and 0,3 of each 5-slot molecule): Cu :186-196
    w = np.arange(40) * 5
    s.constraints = np.stack([np.r_[w, w], np.r_[w + 2, w + 3]], 1).astype(np.int32)
    it = integ(group=g, ngroups=ng)
    t = HostTopology(s, it)
    o = make_oracle(s, g, ng, "TGNH", it)
    assert np.allclose(t.dof()[0], o.dof()[0], rtol=1e-14)
    s.constraints = np.array([[0, 46]], np.int32)               # water of group 0 with a tagged (group 3) water
    assert g[0] != g[46]
    with pytest.raises(TgnhError, match="Temperature group of constrained particles") as e:
        HostTopology(s, integ(group=g, ngroups=ng))
    assert e.value.status == _lib.ERR_GROUP_MISMATCH


def test_unsupported_topologies_fail_loudly():
    s, g, ng = synth.water_box(4)
    bad = s.resid.copy()
    bad[[1, 6]] = bad[[6, 1]]                                   # molecule 0 and 1 interleaved
    s2 = type(s)(mass=s.mass, pair_drude=s.pair_drude, pair_parent=s.pair_parent, resid=bad)
    # (not refused: the reference walks `count` particles from the start of the residue's last run, K :90-91 / Cu :121-124, whoever
    # they belong to; the gather path and the oracle reproduce that walk)
    assert HostTopology(s2, integ()).step_path() == ("gather", "particles of a residue are not contiguous")
    s3 = type(s)(mass=s.mass, pair_drude=np.r_[s.pair_drude, 2], pair_parent=np.r_[s.pair_parent, 0], resid=s.resid)
    with pytest.raises(TgnhError, match="more than one Drude pair"):
        HostTopology(s3, integ())
    m = s.mass.copy(); m[0] = 0.0                               # massless parent: the reference divides by it
    with pytest.raises(TgnhError, match="massless"):
        HostTopology(type(s)(mass=m, pair_drude=s.pair_drude, pair_parent=s.pair_parent, resid=s.resid), integ())
    lone = synth.DrudeSystem(mass=np.array([12.0, 1.0, 0.0]), pair_drude=np.zeros(0, np.int32), pair_parent=np.zeros(0, np.int32),
                             resid=np.array([0, 0, 1], np.int32))                  # molecule 1 = one massless site: v_com = 0/0 in the reference
    with pytest.raises(TgnhError, match="no massive particle"):
        HostTopology(lone, integ())
    HostTopology(lone, integ(com=False))                            # without the COM group nothing divides by that mass


def test_what_the_tiles_cannot_hold_takes_the_gather_path():
    """The reference gathers by arbitrary index (K :171-186) and sizes its bins by G + 2 (K :138-200): a Drude far from its parent,
    pairs that overlap so densely that no tile cut exists, more than 32 temperature groups are not refused -- they step through the
    kernels of tgnh_gather.hip (tgnh_get_step_path says so, and why), with the reference's own index lists as topology."""
    assert HostTopology(synth.water_box(4)[0], integ()).step_path() == ("tiled", "")
    far = synth.DrudeSystem(mass=np.ones(1300), pair_drude=np.array([1200]), pair_parent=np.array([0]), resid=np.zeros(1300, np.int32))
    t = HostTopology(far, integ())                                 # Drude 1200 slots away from its parent
    assert t.step_path()[0] == "gather" and "spans more than one" in t.step_path()[1]
    assert np.array_equal(t.topology(0), np.setdiff1d(np.arange(1300), [0, 1200]))      # normalParticles (Ref :137)
    assert t.topology(1).tolist() == [1200] and t.topology(2).tolist() == [0]
    # pairs nested like onion skins over 700 slots: every cut between slot 1 and 699 goes through some pair
    n = 700
    dense = synth.DrudeSystem(mass=np.ones(n), pair_drude=np.arange(n // 2), pair_parent=n - 1 - np.arange(n // 2), resid=np.zeros(n, np.int32))
    t = HostTopology(dense, integ())
    assert t.step_path()[0] == "gather"
    o = make_oracle(dense, np.zeros(n, np.int32), 1, "TGNH", integ())
    assert np.allclose(t.dof()[0], o.dof()[0], rtol=1e-14)


def test_molecule_longer_than_a_tile_is_tiled_with_a_com_table():
    s, g, ng = synth.polymer_in_water(400, 50)                  # one 1200-slot molecule + 50 waters
    t = HostTopology(s, integ(group=g, ngroups=ng))
    ts = t.topology(7)
    assert np.all(np.diff(ts) <= 512) and ts[-1] == s.num_particles
    tile_of = np.searchsorted(ts, np.arange(s.num_particles), side="right") - 1
    assert np.all(tile_of[s.pair_drude] == tile_of[s.pair_parent])          # pairs are never cut
    assert len(set(tile_of[:1200])) >= 3                                     # the long molecule spans tiles
    for r in range(1, s.num_residues):                                        # the waters do not
        assert len(set(tile_of[s.resid == r])) == 1
    o = make_oracle(s, g, ng, "TGNH", integ(group=g, ngroups=ng))
    assert np.allclose(t.dof()[0], o.dof()[0], rtol=1e-14)


def test_host_only_handle_cannot_launch():
    s, g, ng = synth.water_box(4)
    t = HostTopology(s, integ())
    assert t.lib.tgnh_step_begin(t.h, None) == _lib.ERR_STATE
    assert b"host-only" in t.lib.tgnh_last_error()
    assert t.lib.tgnh_bind_buffers(t.h, None, None, None, None, None) == _lib.ERR_STATE
    # the mailbox exchange needs device memory too, and its calls check their arguments before touching anything
    import ctypes as C
    buf, ptr = C.create_string_buffer(64), C.c_void_p()
    assert t.lib.tgnh_exchange_create(t.h, 2, 0, buf, C.byref(ptr)) == _lib.ERR_STATE
    assert t.lib.tgnh_exchange_attach(t.h, buf.raw * 2) == _lib.ERR_STATE          # no mailbox created
    assert t.lib.tgnh_exchange_attach_pointers(t.h, (C.c_void_p * 2)()) == _lib.ERR_STATE
    assert t.lib.tgnh_exchange_detach(t.h) == 0                                     # nothing attached: a no-op


def test_deferred_rescale_needs_single_group_molecules():
    s, g, ng = synth.water_box(6)
    g = g.copy(); g[2] = 1                                       # H1 of molecule 0 in another group than O
    it = integ(group=g, ngroups=2)
    HostTopology(s, it).close()
    with pytest.raises(TgnhError, match="inside one temperature group"):
        HostTopology(s, integ(group=g, ngroups=2), flags=_lib.FLAG_DEFER_SCALE)


def test_no_drude_pairs():
    """TGNH mode runs without pairs (the Drude thermostat is then inert); dualNH cannot (Ref :181, :472)."""
    s = synth.DrudeSystem(mass=np.array([12.0, 1.0, 1.0, 16.0]), pair_drude=np.zeros(0, np.int32),
                          pair_parent=np.zeros(0, np.int32), resid=np.array([0, 0, 0, 1]))
    t = HostTopology(s, integ())
    assert np.array_equal(t.topology(0), np.arange(4)) and t.dof()[0][0] == 12 - 6 and t.dof()[0][2] == 0
    with pytest.raises(TgnhError, match="at least one Drude pair"):
        HostTopology(s, integ(), mode="dualNH")


def test_dualnh_ignores_temperature_groups():
    """The Reference platform never reads getParticleTempGroup (Ref :426-546: "real" is everything but the Drude motion): an
    integrator that carries groups gives, in dualNH mode, the same handle as one that carries none -- no group index in any
    per-slot word, the same degrees of freedom.  (Used as it came, the array sent kinetic energy into the unused and the Drude
    bins: tools/fuzz_soak.py --modes, round 4.)"""
    s, g, ng = synth.mixed(250, 12)
    assert ng == 4 and g.max() == 3
    with_groups, without = HostTopology(s, integ(group=g, ngroups=ng), mode="dualNH"), HostTopology(s, integ(), mode="dualNH")
    assert not with_groups.topology(3).any()
    for which in (7, 8):                                             # tile starts, packed per-slot words
        assert np.array_equal(with_groups.topology(which), without.topology(which))
    assert not ((with_groups.topology(8).view(np.uint32) >> 2) & 255).any()
    assert np.array_equal(with_groups.dof()[0], without.dof()[0])


def test_up_to_32_temperature_groups():
    s, g, ng = synth.many_groups(100, 8, 32)
    t = HostTopology(s, integ(group=g, ngroups=ng))
    o = make_oracle(s, g, ng, "TGNH", integ(group=g, ngroups=ng))
    assert t.num_thermostats() == 34 and np.allclose(t.dof()[0], o.dof()[0], rtol=1e-14)
    s, g, ng = synth.many_groups(100, 8, 33)                         # K :138-200 sizes its bins by G + 2: no limit there, the gather path here
    t = HostTopology(s, integ(group=g, ngroups=ng))
    o = make_oracle(s, g, ng, "TGNH", integ(group=g, ngroups=ng))
    assert t.step_path() == ("gather", "more than 32 temperature groups")
    assert t.num_thermostats() == 35 and np.allclose(t.dof()[0], o.dof()[0], rtol=1e-14)
    s, g, ng = synth.many_groups(300, 8, 300)
    t = HostTopology(s, integ(group=g, ngroups=ng, chains=3))
    o = make_oracle(s, g, ng, "TGNH", integ(group=g, ngroups=ng, chains=3))
    assert t.num_thermostats() == 302 and np.allclose(t.dof()[0], o.dof()[0], rtol=1e-14)


@pytest.mark.parametrize("seed", range(12))
def test_random_topologies_tile_and_count_like_the_oracle(seed):
    """Ragged inputs: molecules of 1-40 slots in random order of size (a few longer than a tile), Drude pairs anywhere
    inside a molecule (Drude before or after its parent, up to 30 slots apart), massless sites, 1-6 temperature groups
    assigned per molecule, random constraints inside molecules.  Whatever comes: tiles <= 512 slots that cut neither a
    pair nor a molecule that fits a tile, packed words that decode to the input, normal-particle list and degrees of
    freedom equal to the oracle's."""
    from helpers import random_topology
    mass, pd, pp, resid, group, ngroups, cons, sizes, first, rng = random_topology(seed)
    n = len(mass)
    s = synth.DrudeSystem(mass=mass, pair_drude=np.array(pd, np.int32), pair_parent=np.array(pp, np.int32), resid=resid,
                          constraints=np.array(cons, np.int32).reshape(-1, 2), has_cm_motion_remover=bool(seed % 2))
    it = integ(chains=int(rng.integers(1, 5)), group=group, ngroups=ngroups)
    t = HostTopology(s, it)
    o = make_oracle(s, group, ngroups, "TGNH", it)
    assert np.array_equal(t.topology(0), o.normal_particles())
    assert np.allclose(t.dof()[0], o.dof()[0], rtol=1e-13) and np.allclose(t.dof()[1], o.dof()[1], rtol=1e-13)
    ts = t.topology(7)
    assert ts[0] == 0 and ts[-1] == n and np.all(np.diff(ts) > 0) and np.all(np.diff(ts) <= 512)
    tile_of = np.searchsorted(ts, np.arange(n), side="right") - 1
    if len(pd):
        assert np.all(tile_of[s.pair_drude] == tile_of[s.pair_parent])
    for f, sz in zip(first, sizes):
        if sz <= 512:
            assert tile_of[f] == tile_of[f + sz - 1]
    meta = t.topology(8).view(np.uint32)
    off = ((meta >> 10) & 2047).astype(np.int64) - 1024
    assert np.array_equal((np.arange(n) + off)[s.pair_drude], s.pair_parent)
    assert np.array_equal(((meta >> 2) & 255).astype(np.int32), group)
    assert np.array_equal(np.flatnonzero((meta & 3) == 1), np.sort(s.pair_drude))
    assert np.bincount(tile_of, minlength=len(ts) - 1).max() <= 512
