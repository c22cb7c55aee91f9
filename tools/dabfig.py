"""Fast Information Groups for the synthetic ensembles (generator side only, numpy; nothing here is part of the product).

ETSI EN 300 401 clause 5.2 (FIB = 30 bytes of FIGs + CRC16, end marker 0xFF, zero padding), 6.2.1 FIG 0/1 sub-channel organisation
(short form = index into table 8 for UEP, long form = EEP option / level / size), 6.3.1 FIG 0/2 services and their components,
6.3.2 FIG 0/3 packet-mode components, 6.2.2 FIG 0/14 FEC scheme, 6.3.6 FIG 0/13 user applications, 6.4 FIG 0/0 ensemble
information (CIF counter), 8.1.3.2 FIG 0/9 country / LTO / international table, 8.1.5 FIG 0/17 programme type, 8.1.13 / 8.1.14
FIG 1/0, 1/1, 1/5 labels.  These are the groups the reference's FIG_Processor parses (/root/reference/src/dab/fic/fig_processor.cpp:94-160
ProcessFIB, :268-560 FIG 0/0..0/3, :1034-1080 FIG 0/9, :1186-1274 FIG 0/13, 0/14, :1277-1345 FIG 0/17, :1618-1790 FIG 1/x) and that its
Radio_FIG_Handler / DAB_Database_Updater turn into the database from which basic_radio creates its MSC decoders lazily
(/root/reference/src/basic_radio/basic_radio.cpp:83-154): a synthetic ensemble whose FIBs carry them can be consumed by the
reference's own control code downstream of the bits-out boundary, which random FIB bytes cannot.

`describe(layout)` assigns identifiers, services and components to a multiplex; `Carousel` packs the groups into FIBs frame by
frame -- the sub-channel organisation and the service lists are spread over the first frames so that the database entries complete
(and the decoders appear) at DIFFERENT frames, then everything repeats like a transmitter's carousel."""
import numpy as np

FIB_DATA_BYTES = 30
EEP_A, EEP_B = 0, 1


def _u8(*vals):
    return bytes(int(v) & 0xFF for v in vals)


def label16(text):
    b = text.encode("ascii")[:16]
    return b + b" " * (16 - len(b))


def describe(layout, seed=0, packet_sub=None, stream_data_sub=None, orphan_sub=None, eid=0xE1C5, ecc=0xE1):
    """layout: list of (start CU, length CU, is_uep, uep_index, eep_level, eep_type) or dicts with those keys (tools/dabsynth.py layouts).
    Returns a dict: ensemble, subchannels (with their identifiers, which are NOT the list positions) and services.
    Every sub-channel gets one service with one primary component: UEP -> MPEG audio (ASCTy 0), EEP -> DAB+ audio (ASCTy 63);
    packet_sub -> a packet-mode data service (32-bit SId, FIG 0/3 + 0/13 + 0/14 complete it), stream_data_sub -> a stream-mode data
    component (basic_radio.cpp creates no decoder for one), orphan_sub -> organised in FIG 0/1 but referenced by no service."""
    rng = np.random.default_rng(seed + 4711)
    subs = []
    for k, d in enumerate(layout):
        if isinstance(d, dict):
            t = (d["start"], d["length"], d["is_uep"], d["uep_index"], d["eep_level"], d["eep_type"])
        elif hasattr(d, "start_address"):
            t = (d.start_address, d.length, int(d.is_uep), d.uep_index, d.eep_level, d.eep_type)
        else:
            t = tuple(d)
        subs.append(dict(id=(5 * k + 3) % 64, start=int(t[0]), length=int(t[1]), is_uep=int(t[2]), uep_index=int(t[3]), eep_level=int(t[4]), eep_type=int(t[5]),
                         fec=None, index=k))
    services = []
    for k, s in enumerate(subs):
        if k == orphan_sub:
            continue
        name = "SVC %02d %s" % (k, "".join(chr(65 + int(c)) for c in rng.integers(0, 26, 5)))
        if k == packet_sub:
            s["fec"] = 1
            services.append(dict(sid=(ecc << 24) | (0xE << 20) | (0x40000 + k), sid32=True, label=name, flag=0xFF00, pty=None,
                                 comp=dict(kind="packet", scid=0x200 + k, dscty=60, packet_addr=0x120 + k, user_app=0x002, sub=s["id"])))
        elif k == stream_data_sub:
            services.append(dict(sid=0xE000 | (0x800 + k), sid32=False, label=name, flag=0xF0F0, pty=None,
                                 comp=dict(kind="stream_data", dscty=5, sub=s["id"])))
        else:
            services.append(dict(sid=0xE000 | (0x100 + 7 * k), sid32=False, label=name, flag=0xFF00 >> (k % 5), pty=int(1 + k % 29),
                                 comp=dict(kind="audio", ascty=0 if s["is_uep"] else 63, sub=s["id"])))
    return dict(eid=eid, ecc=ecc, lto=0x02, inter_table=1, label="GRAFT MUX %03d" % (seed % 1000), flag=0xFC00, subchannels=subs, services=services)


# ---- the groups: every function returns (type, key, entry bytes, tag); entries with equal (type, key) may share one FIG;
# tag = (what, identifier) tells a test which database entry the group contributes to ----
def fig0(ext, body, pd=0, tag=None):
    return (0, (ext, pd), bytes(body), tag)


def fig_0_0(desc, cif_count):
    hi, lo = (cif_count // 250) % 20, cif_count % 250
    return fig0(0, _u8(desc["eid"] >> 8, desc["eid"], hi & 0x1F, lo))


def fig_0_1(s):
    head = _u8((s["id"] << 2) | (s["start"] >> 8), s["start"])
    if s["is_uep"]:
        return fig0(1, head + _u8(s["uep_index"] & 0x3F), tag=("organisation", s["id"]))           # short form, table switch 0
    return fig0(1, head + _u8(0x80 | ((s["eep_type"] & 7) << 4) | ((s["eep_level"] & 3) << 2) | (s["length"] >> 8), s["length"]), tag=("organisation", s["id"]))


def _sid_bytes(svc):
    n = 4 if svc["sid32"] else 2
    return svc["sid"].to_bytes(n, "big")


def fig_0_2(svc):
    c = svc["comp"]
    if c["kind"] == "audio":
        comp = _u8((0 << 6) | c["ascty"], (c["sub"] << 2) | 2)
    elif c["kind"] == "stream_data":
        comp = _u8((1 << 6) | c["dscty"], (c["sub"] << 2) | 2)
    else:
        comp = _u8((3 << 6) | (c["scid"] >> 6), ((c["scid"] & 0x3F) << 2) | 2)
    return fig0(2, _sid_bytes(svc) + _u8(1) + comp, pd=int(svc["sid32"]), tag=("component", c["sub"]))


def fig_0_3(svc):
    c = svc["comp"]
    return fig0(3, _u8(c["scid"] >> 4, (c["scid"] & 0xF) << 4, c["dscty"] & 0x3F, (c["sub"] << 2) | (c["packet_addr"] >> 8), c["packet_addr"]),
                tag=("packet", c["sub"]))


def fig_0_9(desc):
    return fig0(9, _u8(desc["lto"] & 0x3F, desc["ecc"], desc["inter_table"]))


def fig_0_13(svc):
    c = svc["comp"]
    return fig0(13, _sid_bytes(svc) + _u8((0 << 4) | 1, c["user_app"] >> 3, (c["user_app"] & 7) << 5), pd=int(svc["sid32"]), tag=("user_app", c["sub"]))


def fig_0_14(s):
    return fig0(14, _u8((s["id"] << 2) | (s["fec"] & 3)), tag=("fec", s["id"]))


def fig_0_17(svc):
    return fig0(17, _sid_bytes(svc) + _u8(0, svc["pty"] & 0x1F))


def fig_1(ext, ident, text, flag):
    return (1, (ext, 0), _u8(ext) + ident + label16(text) + _u8(flag >> 8, flag), None)              # charset 0 (EBU Latin); one label per FIG


def pack_fib(groups):
    """as many of `groups` (in order) as fit into one FIB -> (30 data bytes, number consumed).  Consecutive FIG type 0 entries of the
    same extension and P/D share one FIG (header + descriptor byte + entries, data field <= 29 bytes)."""
    out, used = bytearray(), 0
    cur = None                                              # [type, key, body bytearray] of the FIG under construction
    def flush():
        nonlocal cur
        if cur is not None:
            t, key, body = cur
            data = (_u8(key[0] | (key[1] << 5)) + bytes(body)) if t == 0 else bytes(body)
            out.extend(_u8((t << 5) | len(data)) + data)
            cur = None
    for t, key, entry, _tag in groups:
        have = len(out) + (0 if cur is None else (2 if cur[0] == 0 else 1) + len(cur[2]))
        if cur is not None and t == 0 and cur[0] == 0 and cur[1] == key and have + len(entry) <= FIB_DATA_BYTES and 1 + len(cur[2]) + len(entry) <= 29:
            cur[2].extend(entry)
        else:
            need = (2 if t == 0 else 1) + len(entry)
            if have + need > FIB_DATA_BYTES:
                break
            flush()
            cur = [t, key, bytearray(entry)]
        used += 1
    flush()
    assert len(out) <= FIB_DATA_BYTES
    if len(out) < FIB_DATA_BYTES:
        out.extend(b"\xFF" + b"\x00" * (FIB_DATA_BYTES - len(out) - 1))
    return bytes(out), used


class Carousel:
    """frame f -> [4][3][30] FIB data bytes.  FIB 0 of every frame opens with FIG 0/0 carrying the CIF count of the frame's first CIF.
    `spread` sub-channels are organised per frame (FIG 0/1) and each one's service (FIG 0/2, and 0/3 / 0/13 / 0/14 for a packet-mode
    one) follows ONE FRAME LATER, so sub-channel k's entry completes in frame k // spread + 1; FIG 0/9, labels and programme types
    follow; after that the whole list repeats from the start (a carousel), `per_frame` groups per frame."""

    def __init__(self, desc, spread=3, per_frame=10):
        self.desc, self.spread = desc, spread
        subs = desc["subchannels"]
        by_sub = {svc["comp"]["sub"]: svc for svc in desc["services"]}
        n_waves = (len(subs) + spread - 1) // spread
        self.intro = []                                      # the first n_waves + 1 frames, group lists
        for w in range(n_waves + 1):
            g = []
            for s in subs[w * spread:(w + 1) * spread]:
                g.append(fig_0_1(s))
            if w >= 1:
                for s in subs[(w - 1) * spread:w * spread]:
                    svc = by_sub.get(s["id"])
                    if svc is None:
                        continue
                    g.append(fig_0_2(svc))
                    if svc["comp"]["kind"] == "packet":
                        g += [fig_0_3(svc), fig_0_13(svc), fig_0_14(s)]
            self.intro.append(g)
        self.si = [fig_0_9(desc), fig_1(0, _u8(desc["eid"] >> 8, desc["eid"]), desc["label"], desc["flag"])]
        for svc in desc["services"]:
            self.si.append(fig_1(5 if svc["sid32"] else 1, _sid_bytes(svc), svc["label"], svc["flag"]))
            if svc["pty"] is not None:
                self.si.append(fig_0_17(svc))
        self.rate = per_frame
        self.loop = [g for wave in self.intro for g in wave] + self.si
        self.loop_pos = 0
        self.si_pos = 0
        self.log = []                                        # per frame: the tags of the groups it carried

    def frame(self, f):
        """must be called for f = 0, 1, 2, ... in order"""
        groups = [fig_0_0(self.desc, 4 * f)]
        if f < len(self.intro):
            groups += self.intro[f]
        elif self.si_pos < len(self.si):
            take = self.si[self.si_pos:self.si_pos + self.rate]
            self.si_pos += len(take)
            groups += take
        else:
            for _ in range(self.rate):
                groups.append(self.loop[self.loop_pos % len(self.loop)])
                self.loop_pos += 1
        fibs = np.zeros((4, 3, FIB_DATA_BYTES), np.uint8)
        pos = 0
        for k in range(12):
            data, used = pack_fib(groups[pos:])
            pos += used
            fibs[k // 3, k % 3] = np.frombuffer(data, np.uint8)
        self.log.append([g[3] for g in groups if g[3] is not None])
        assert pos == len(groups), "frame %d: %d of %d groups did not fit into 12 FIBs" % (f, len(groups) - pos, len(groups))
        return fibs

    def frames(self, n):
        return np.stack([self.frame(f) for f in range(n)])

    def decoder_after_frame(self, k, received_frames):
        """the frame after whose FIC sub-channel index k and its service component are both complete, i.e. in whose UpdateAfterProcessing
        basic_radio.cpp:83-154 creates the decoder, given the set of frames whose FIBs were received (all twelve of them) --
        None: never (no component, a stream-mode data component, or not within the frames generated so far).
        Sub-channel complete = organisation seen (dab_database_updater.cpp:222-227); audio component complete = FIG 0/2 seen; packet-mode
        component = FIG 0/2, then 0/3 and 0/13 (:161-168), and the decoder also needs the FEC scheme of FIG 0/14 (basic_radio.cpp:144)."""
        s = self.desc["subchannels"][k]
        svc = [v for v in self.desc["services"] if v["comp"]["sub"] == s["id"]]
        if not svc or svc[0]["comp"]["kind"] == "stream_data":
            return None
        need = {"organisation", "component"} | ({"packet", "user_app", "fec"} if svc[0]["comp"]["kind"] == "packet" else set())
        have = set()
        for f, tags in enumerate(self.log):
            if f not in received_frames:
                continue
            for what, ident in tags:
                if ident == s["id"] and (what != "packet" or "component" in have):        # FIG 0/3 before FIG 0/2 finds no component to update
                    have.add(what)
            if need <= have:
                return f
        return None


def repeating_frames(desc, nf):
    """[nf][4][3][30] FIB data for a transmission that REPEATS every nf frames (tools/dabsynth.py::Multiplex): every frame opens with FIG 0/0, the
    sub-channel organisation, the service lists and then -- as far as 12 FIBs per frame hold them -- FIG 0/9, programme types and labels are dealt out
    over the nf frames, so that after one repetition a receiver has seen every group.  Returns (frames, number of groups that did not fit)."""
    car = Carousel(desc)
    mci = [g for wave in car.intro for g in wave]
    groups = [g for g in mci if g[1][0] == 1] + [g for g in mci if g[1][0] != 1] + car.si      # organisation, then services, then SI
    per_frame = [[fig_0_0(desc, 4 * f)] for f in range(nf)]
    for k, g in enumerate(groups):
        per_frame[k % nf].append(g)
    out = np.zeros((nf, 4, 3, FIB_DATA_BYTES), np.uint8)
    dropped = 0
    for f in range(nf):
        pos = 0
        for k in range(12):
            data, used = pack_fib(per_frame[f][pos:])
            pos += used
            out[f, k // 3, k % 3] = np.frombuffer(data, np.uint8)
        dropped += len(per_frame[f]) - pos                     # (only trailing SI groups can be left over: the MCI comes first)
    return out, dropped


def expected_database(desc):
    """what the reference's database must hold once every group has been seen: a canonical text, line for line what
    tests/cpp/ref_callers_driver.cpp dumps from DAB_Database (sorted)"""
    lines = ["ensemble id=%04X ecc=%02X lto=%d inter_table=%d label=[%s]" % (desc["eid"], desc["ecc"], 5 * (desc["lto"] & 0x1F) * (-1 if desc["lto"] & 0x20 else 1),
                                                                              desc["inter_table"], label16(desc["label"]).decode())]
    for s in sorted(desc["subchannels"], key=lambda s: s["id"]):
        from_table = s["length"]
        if s["is_uep"]:
            lines.append("subchannel id=%d start=%d length=%d uep index=%d fec=%s complete=1" % (s["id"], s["start"], from_table, s["uep_index"], "none"))
        else:
            lines.append("subchannel id=%d start=%d length=%d eep level=%d type=%s fec=%s complete=1" % (
                s["id"], s["start"], s["length"], s["eep_level"], "AB"[s["eep_type"]], "none" if s["fec"] is None else str(s["fec"])))
    for svc in sorted(desc["services"], key=lambda v: v["sid"]):
        lines.append("service id=%X bits=%d label=[%s] pty=%d" % (svc["sid"], 32 if svc["sid32"] else 16, label16(svc["label"]).decode(), svc["pty"] or 0))
    for svc in sorted(desc["services"], key=lambda v: v["sid"]):
        c = svc["comp"]
        if c["kind"] == "audio":
            lines.append("component service=%X scids=0 subchannel=%d mode=stream_audio audio=%d complete=1" % (svc["sid"], c["sub"], c["ascty"]))
        elif c["kind"] == "stream_data":
            lines.append("component service=%X scids=0 subchannel=%d mode=stream_data data=%d complete=%d" % (svc["sid"], c["sub"], c["dscty"], 1))
        else:
            lines.append("component service=%X scids=0 subchannel=%d mode=packet_data data=%d scid=%d packet_addr=%d apps=%d complete=1" % (
                svc["sid"], c["sub"], c["dscty"], c["scid"], c["packet_addr"], c["user_app"]))
    return lines
