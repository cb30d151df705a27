"""The device's f64 → JSON text routine (loro_amd/csrc/lm_f64.h, compiled for the host inside tests/emu) against
the oracle (std::to_chars shortest round-trip digits + ryu's pretty layout) and against known serde_json outputs."""
import ctypes, random, struct

import _oracle, _emu


def _fns():
    e = _emu.binding().lib
    e.lmemu_f64_json.restype = ctypes.c_int
    e.lmemu_f64_json.argtypes = [ctypes.c_uint64, ctypes.c_char_p]
    o = _oracle.lib()
    o.lo_json_f64.restype = ctypes.c_int
    o.lo_json_f64.argtypes = [ctypes.c_double, ctypes.c_char_p]
    b1, b2 = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)

    def dev(bits):
        n = e.lmemu_f64_json(bits, b1)
        return b1.raw[:n]

    def ora(bits):
        n = o.lo_json_f64(struct.unpack("<d", struct.pack("<Q", bits))[0], b2)
        return b2.raw[:n]
    return dev, ora


def _bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def test_known_serde_json_outputs():
    dev, _ = _fns()
    for x, want in [(1.5, b"1.5"), (0.0, b"0.0"), (-0.0, b"-0.0"), (1.0, b"1.0"), (100.0, b"100.0"), (0.1, b"0.1"), (1e16, b"1e16"),
                    (1e15, b"1000000000000000.0"), (123456789012345680.0, b"1.2345678901234568e17"), (1e-5, b"0.00001"), (1e-6, b"1e-6"),
                    (1e-7, b"1e-7"), (5e-324, b"5e-324"), (1.7976931348623157e308, b"1.7976931348623157e308"), (2.2250738585072014e-308, b"2.2250738585072014e-308"),
                    (0.3, b"0.3"), (2 / 3, b"0.6666666666666666"), (-2.5e-3, b"-0.0025"), (1e21, b"1e21"), (9007199254740993.0, b"9007199254740992.0"),
                    (float("inf"), b"null"), (float("nan"), b"null")]:
        assert dev(_bits(x)) == want, (x, dev(_bits(x)))
        if x == x and abs(x) != float("inf"):
            assert float(dev(_bits(x))) == x


def test_random_doubles_match_the_oracle_and_round_trip():
    dev, ora = _fns()
    rng = random.Random(5)
    cases = []
    for _ in range(60000):
        cases.append(rng.getrandbits(64))                                   # uniform bit patterns (all exponents)
    for _ in range(20000):
        cases.append(_bits(rng.uniform(-1e6, 1e6)))
        cases.append(_bits(round(rng.uniform(-1e4, 1e4), rng.randint(0, 6))))  # short decimals
        cases.append(_bits(float(rng.randint(-10**17, 10**17))))
    for _ in range(60000):                                                  # every binary exponent the two-register path takes (-121 <= e2 <= 60) and its edges
        cases.append((rng.getrandbits(1) << 63) | (rng.randint(1075 - 124, 1075 + 63) << 52) | rng.getrandbits(52))
        cases.append(_bits(rng.random() * 10.0 ** rng.randint(-22, 35)))
    for e in range(1075 - 124, 1075 + 64):                                   # … its powers of two, their neighbours, all-ones fractions
        cases += [e << 52, (e << 52) + 1, (e << 52) - 1, (e << 52) | ((1 << 52) - 1), (e << 52) | (1 << 51)]
    for d in range(1, 18):                                                  # decimals of every length around the powers of ten of that range
        for ex in range(-21, 34):
            cases.append(_bits(float("%de%d" % (rng.randint(10 ** (d - 1), 10 ** d - 1), ex - d))))
    for e in range(-1074, 1024, 7):                                         # powers of two (uneven neighbours) and ±1 ulp
        b = _bits(2.0 ** e) if e > -1023 else 1 << (e + 1074)
        cases += [b, b + 1, max(b - 1, 1)]
    for bits in cases:
        if (bits >> 52) & 0x7FF == 0x7FF:
            assert dev(bits) == b"null"
            continue
        d = dev(bits)
        assert d == ora(bits), (hex(bits), d, ora(bits))
        assert _bits(float(d)) == bits or (bits << 1) == 0
