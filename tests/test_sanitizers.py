"""The module's text grammar (csrc/pocs_command.hpp: host only, no GPU) fuzzed under AddressSanitizer + UBSan.

The reference's handlers have real undefined behaviour at exactly this boundary: `setAlphas` copies every remaining
token into a 1 x 4 matrix (mcsimplugin.cpp:176-184 + MCSimulator.h:143,226-228: a fifth token writes out of bounds),
`setLandmarks` / `setTrajectory` / `setOdometry` loop to counts that earlier commands may never have set (:83-113,
:148-166), eight handlers fall off the end of a bool function (:83-172).  tests/command_fuzz.cpp -- always compiled
with -fsanitize=address,undefined -- feeds lines to the same parser pocs_send_command dispatches on; a finding aborts it.
  * lines of the reference's grammar (hypothesis): the parse is what a Python restatement of the grammar says;
  * the overflow cases by name; empty tails; arbitrary bytes: a verdict for every line, never a crash.
`make -C tests sanitize` runs the whole CPU suite against sanitizer builds of the other host-side pieces as well."""
import math
import subprocess
from pathlib import Path

import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

ROOT = Path(__file__).resolve().parents[1]
HERE = Path(__file__).resolve().parent
E_ARG, E_ORDER, E_UNKNOWN = -1, -2, -5

# command -> (id in pocs_cmd::Id order, what it takes)
IDS = ["MyCommand", "ArmaCommand", "help", "setAlphas", "setQ", "setNumLandmarks", "setLandmarks", "setNumParticles",
       "setInitialCovariance", "setPathLength", "setTrajectory", "setOdometry", "runSimulation", "setNumGaussians",
       "runGMMEstimation", "setNumGMMSamples", "setSeed", "setFootprint", "addObstacle", "clearObstacles", "setBatch",
       "setRunAhead"]
UNKNOWN = len(IDS)
NO_TOKENS = {"MyCommand", "ArmaCommand", "help", "clearObstacles", "runSimulation", "runGMMEstimation"}
ONE_INT = {"setNumLandmarks", "setNumParticles", "setPathLength", "setNumGaussians", "setNumGMMSamples", "setBatch", "setRunAhead"}
FIXED = {"setQ": 1, "setInitialCovariance": 9, "setFootprint": 4, "addObstacle": 5}


@pytest.fixture(scope="module")
def fuzz_exe():
    src, exe = HERE / "command_fuzz.cpp", HERE / "_command_fuzz"
    hdr = ROOT / "probability-of-collision-for-safe-planning_amd" / "csrc" / "pocs_command.hpp"
    if not exe.exists() or exe.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Werror", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=undefined", str(src), "-o", str(exe)], check=True)
    return exe


def run_records(exe, records):
    """records: [(num_landmarks, path_length, line bytes)] -> [(id, err, values, n, seed, name, msg)]"""
    blob = b"".join(b"%d %d %d " % (nl, W, len(line)) + line for nl, W, line in records)
    out = subprocess.run([str(exe)], input=blob, capture_output=True, check=False,
                         env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"})
    assert out.returncode == 0, out.stderr.decode("latin1")[-3000:]           # a sanitizer finding aborts the driver
    lines = out.stdout.decode("latin1").split("\n")
    res, i = [], 0
    for _ in records:
        head, name, msg = lines[i].split("|", 2)
        cid, err, nv, n, seed = head.split()
        vals = [float(v) for v in lines[i + 1:i + 1 + int(nv)]]
        res.append((int(cid), int(err), vals, int(n), int(seed), name, msg))
        i += 1 + int(nv)
    return res


def model(nl, W, name, tokens, seed_text=None):
    """The grammar restated: (id, err, values) for a line `name tok tok ...` whose tokens are decimal numbers or junk."""
    cid = IDS.index(name) if name in IDS else UNKNOWN
    if cid == UNKNOWN:
        return cid, E_UNKNOWN, None
    if name in NO_TOKENS:
        return cid, 0, []
    if name == "setSeed":
        return cid, (0 if seed_text is not None else E_ARG), None
    if name == "setLandmarks" and nl < 0:
        return cid, E_ORDER, None
    if name in ("setTrajectory", "setOdometry") and W < 1:
        return cid, E_ORDER, None
    vals = []
    for t in tokens:
        try:
            if "_" in t:                                            # (Python's float() takes digit separators, strtod does not)
                raise ValueError(t)
            vals.append(float(t))
        except ValueError:
            return cid, E_ARG, None                                 # malformed number
    if name == "setAlphas":
        return cid, (0 if 1 <= len(vals) <= 4 else E_ARG), vals
    want = 1 if name in ONE_INT else FIXED.get(name)
    if name == "setLandmarks":
        want = 2 * nl
    if name == "setTrajectory":
        want = 3 * W
    if name == "setOdometry":
        want = 3 * (W - 1)
    if len(vals) != want:
        return cid, E_ARG, vals
    if name in ONE_INT and not (vals[0] == math.floor(vals[0]) and abs(vals[0]) < 9.0e15):
        return cid, E_ARG, vals
    return cid, 0, vals


numbers = st.one_of(st.floats(allow_nan=False, allow_infinity=False, width=64).map(repr),
                    st.integers(-10 ** 6, 10 ** 6).map(str),
                    st.sampled_from(["0", "-0.0", "1e-300", "2.5e+300", "007", ".5", "5.", "+3"]))
junk = st.sampled_from(["abc", "1.2.3", "--5", "1e", "e5", "0x", "1,2", "3;", "nan()x", "1_0", "\x7f", "@"])
token = st.one_of(numbers, numbers, numbers, junk)
blank = st.sampled_from([" ", "  ", "\t", " \t ", "\n", "\r\n"])
names = st.one_of(st.sampled_from(IDS), st.sampled_from(IDS), st.sampled_from(["", "setalphas", "SetQ", "run", "setAlphas1", "help!", "\x00x"]))


@st.composite
def command_lines(draw):
    nl = draw(st.integers(-1, 6))
    W = draw(st.integers(-1, 5))
    name = draw(names)
    # bias the count towards what the command takes (and one off either way): the interesting edge
    want = {"setLandmarks": 2 * max(nl, 0), "setTrajectory": 3 * max(W, 0), "setOdometry": 3 * max(W - 1, 0), "setAlphas": 4}.get(
        name, 1 if name in ONE_INT else FIXED.get(name, 0))
    count = draw(st.one_of(st.just(want), st.just(want + 1), st.just(max(want - 1, 0)), st.integers(0, 20)))
    toks = [draw(token) for _ in range(count)]
    lead = draw(st.sampled_from(["", " ", "\t "]))
    seps = [draw(blank) for _ in range(count + 1)]
    tail = draw(st.sampled_from(["", " ", "\n", " \t\n"]))
    line = lead + name + "".join(s + t for s, t in zip(seps, toks)) + tail
    if "\x00" in name:
        name = name.split("\x00")[0]                                 # the line ends at its first NUL
        toks = []
        line = lead + name
    return nl, W, name, toks, line


@settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(st.lists(command_lines(), min_size=1, max_size=40))
def test_grammar_against_its_restatement(fuzz_exe, cases):
    recs = [(nl, W, line.encode("latin1")) for nl, W, _, _, line in cases]
    got = run_records(fuzz_exe, recs)
    for (nl, W, name, toks, line), (cid, err, vals, n, seed, gname, msg) in zip(cases, got):
        if name == "setSeed":
            continue                                                 # (its own test below: strtoull's grammar)
        want_id, want_err, want_vals = model(nl, W, name, toks)
        assert (cid, err) == (want_id, want_err), (line, cid, err, msg, want_id, want_err)
        if err == 0 and want_vals is not None:
            assert vals == want_vals, (line, vals, want_vals)
            if name in ONE_INT:
                assert n == int(want_vals[0])
        if err != 0:
            assert msg                                               # every refusal says why


def test_the_reference_s_own_overflows_are_refusals(fuzz_exe):
    """The cases mcsimplugin.cpp handles by undefined behaviour, one by one."""
    cases = [
        (8, 56, b"setAlphas 1 2 3 4 5", E_ARG),                      # :176-184: a fifth token wrote past the 1 x 4 matrix
        (8, 56, b"setAlphas " + b"1 " * 4096, E_ARG),
        (8, 56, b"setAlphas", E_ARG),                                # empty tail
        (8, 56, b"setAlphas   \n", E_ARG),
        (-1, 56, b"setLandmarks 1 2 3 4", E_ORDER),                  # :148-166 before setNumLandmarks: the loop bound was never set
        (2, 56, b"setLandmarks 1 2 3", E_ARG),                       # one token short: operator>> left the rest uninitialised
        (2, 56, b"setLandmarks 1 2 3 4 5", E_ARG),
        (8, -1, b"setTrajectory 1 2 3", E_ORDER),                    # :83-97 before setPathLength
        (8, 1, b"setOdometry", 0),                                   # W = 1: zero steps, zero tokens -- a legal empty tail
        (8, 2, b"setOdometry 1 2", E_ARG),
        (8, 56, b"setQ", E_ARG), (8, 56, b"setQ 0.04 1", E_ARG), (8, 56, b"setQ 0.04x", E_ARG),
        (8, 56, b"setNumParticles 1e3", 0), (8, 56, b"setNumParticles 2.5", E_ARG), (8, 56, b"setNumParticles 1e300", E_ARG),
        (8, 56, b"setNumParticles nan", E_ARG), (8, 56, b"setNumParticles inf", E_ARG),
        (8, 56, b"runGMMEstimation", 0), (8, 56, b"runGMMEstimation   ", 0), (8, 56, b"runSimulation trailing junk", 0),
        (8, 56, b"", E_UNKNOWN), (8, 56, b"   ", E_UNKNOWN), (8, 56, b"\n", E_UNKNOWN), (8, 56, b"mycommand", E_UNKNOWN),
        (8, 56, b"setSeed", E_ARG), (8, 56, b"setSeed x", E_ARG), (8, 56, b"setSeed 0x5EED0001", 0), (8, 56, b"setSeed 18446744073709551615", 0),
        (8, 56, b"setSeed 99999999999999999999999", 0),               # strtoull saturates: a seed is a seed
    ]
    got = run_records(fuzz_exe, [(nl, W, line) for nl, W, line, _ in cases])
    for (nl, W, line, want), (cid, err, vals, n, seed, name, msg) in zip(cases, got):
        assert err == want, (line[:40], err, msg)
    assert got[27][4] == 0x5EED0001 and got[28][4] == 2 ** 64 - 1 and got[29][4] == 2 ** 64 - 1
    assert got[13][3] == 1000


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(st.lists(st.tuples(st.integers(-3, 40), st.integers(-3, 600), st.binary(min_size=0, max_size=400)), min_size=1, max_size=30))
def test_arbitrary_bytes_get_a_verdict(fuzz_exe, recs):
    """Any bytes at all (NULs, high bytes, huge exponents, no terminator): a verdict for each, no sanitizer finding."""
    got = run_records(fuzz_exe, recs)
    for (nl, W, line), (cid, err, vals, n, seed, name, msg) in zip(recs, got):
        assert 0 <= cid <= UNKNOWN and err in (0, E_ARG, E_ORDER, E_UNKNOWN)
        assert (cid == UNKNOWN) == (err == E_UNKNOWN)
        if err == 0 and IDS[cid] == "setLandmarks":
            assert len(vals) == 2 * nl
        if err == 0 and IDS[cid] == "setTrajectory":
            assert len(vals) == 3 * W
        if err == 0 and IDS[cid] == "setAlphas":
            assert 1 <= len(vals) <= 4
