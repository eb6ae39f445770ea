#!/usr/bin/env python3
"""Generate tests/golden/surface_cases.json from the REAL reference readers (oracle/_ref): what
cpp/include/pgm_io.hpp's read_pgm and minijson + the config getters of templering_sfm.cpp (T:65-106)
return or throw for a list of inputs -- header comments, maxval != 255, short files, malformed and
duplicate-key JSON.  The cases are data (input bytes + the reference's outcome); the product's
readers are held to them in tests/test_host_math.py.  Run in the build container only:

    python tests/golden/make_surface_golden.py
"""
from __future__ import annotations

import base64
import ctypes
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers as H  # noqa: E402


def pgm_inputs():
    px = bytes(range(12))
    yield "plain", b"P5\n4 3\n255\n" + px
    yield "comment_after_magic_same_line", b"P5# made by x\n4 3\n255\n" + px
    yield "comment_after_newline", b"P5\n# made by x\n4 3\n255\n" + px          # NOT skipped by the reference (peek sees '\n')
    yield "comment_after_width", b"P5 4# w\n3\n255\n" + px
    yield "comment_after_height", b"P5 4 3# h\n255\n" + px
    yield "two_comments", b"P5# a\n# b\n4 3\n255\n" + px
    yield "maxval_65535", b"P5\n4 3\n65535\n" + px * 2
    yield "maxval_15", b"P5\n4 3\n15\n" + px
    yield "p2_ascii", b"P2\n4 3\n255\n" + b"1 2 3 4 5 6 7 8 9 10 11 12\n"
    yield "short_pixels", b"P5\n4 3\n255\n" + px[:7]
    yield "no_separator_byte", b"P5\n4 3\n255"
    yield "crlf_header", b"P5\r\n4 3\r\n255\n" + px
    yield "tabs", b"P5\t4\t3\t255\t" + px
    yield "extra_bytes_after_pixels", b"P5\n4 3\n255\n" + px + b"tail"
    yield "separator_is_pixel", b"P5\n4 3\n255" + bytes([9]) + px                  # get() eats exactly one byte, whatever it is
    yield "non_numeric_width", b"P5\nab 3\n255\n" + px
    yield "empty_file", b""
    yield "zero_size", b"P5\n0 0\n255\n"
    yield "plus_sign", b"P5\n+4 3\n255\n" + px


def json_inputs():
    base = '{"common": {"klt": {"max_tracks": 500, "quality": 0.02}, "system": {"frames": 7}, "outputs": {"export_geometry": "mesh"}}, ' \
           '"cpp": {"klt": {"max_tracks": 333.6}, "ba": {"lambda": 1e-2}}}'
    q = [("klt", "max_tracks", 0), ("klt", "max_tracks", 1), ("klt", "quality", 1), ("system", "frames", 0), ("outputs", "export_geometry", 2),
         ("ba", "lambda", 1), ("ba", "window", 0), ("klt", "quality", 2), ("outputs", "export_geometry", 0)]
    for s, k, kind in q:
        yield f"lookup_{s}_{k}_{kind}", base, s, k, kind
    docs = {
        "dup_key_first_wins": '{"common": {"klt": {"iters": 3, "iters": 9}}}',
        "dup_section_first_wins": '{"common": {"klt": {"iters": 3}}, "common": {"klt": {"iters": 9}}}',
        "negative_half_rounds_away": '{"cpp": {"klt": {"iters": -2.5}}}',
        "exponent": '{"cpp": {"klt": {"iters": 1.2e1}}}',
        "unicode_escape": '{"common": {"klt": {"iters": "a\\u00e9\\u20ac\\n"}}}',
        "bool_is_not_a_number": '{"common": {"klt": {"iters": true}}}',
        "null_value": '{"common": {"klt": {"iters": null}}}',
        "array_value": '{"common": {"klt": {"iters": [1, 2]}}}',
        "section_is_not_object": '{"common": {"klt": 5}}',
        "whitespace": ' \t\n{ "common" :\n{ "klt" : { "iters" : 4 } } }\n ',
        "trailing_comma_object": '{"common": {"klt": {"iters": 4,}}}',
        "trailing_comma_array": '{"common": {"klt": {"iters": [1,]}}}',
        "missing_colon": '{"common" {"klt": {"iters": 4}}}',
        "missing_comma": '{"common": {"klt": {"iters": 4 "x": 1}}}',
        "unterminated_string": '{"common": {"klt": {"iters": "abc',
        "bad_escape": '{"common": {"klt": {"iters": "a\\q"}}}',
        "bad_hex": '{"common": {"klt": {"iters": "\\u12g4"}}}',
        "short_hex": '{"common": {"klt": {"iters": "\\u12',
        "leading_zero": '{"common": {"klt": {"iters": 012}}}',
        "bad_fraction": '{"common": {"klt": {"iters": 1.}}}',
        "bad_exponent": '{"common": {"klt": {"iters": 1e+}}}',
        "lone_minus": '{"common": {"klt": {"iters": -}}}',
        "bad_literal": '{"common": {"klt": {"iters": nul}}}',
        "bad_true": '{"common": {"klt": {"iters": tru}}}',
        "unexpected_char": '{"common": {"klt": {"iters": @}}}',
        "trailing_garbage": '{"common": {"klt": {"iters": 4}}} x',
        "empty_text": '',
        "only_space": '   ',
        "key_not_string": '{common: 1}',
        "top_level_array": '[1, 2, 3]',
        "top_level_number": '42',
        "unclosed_object": '{"common": {"klt": {"iters": 4}}',
    }
    for name, text in docs.items():
        for kind in (0, 2):
            yield f"{name}_{kind}", text, "klt", "iters", kind


CLI_FILES = {
    "data/templeRing/templeR_par.txt": "1\ntempleR0001.png 1520.4 0 302.32 0 1525.9 246.87 0 0 1 1 0 0 0 1 0 0 0 1 0 0 0.6\n",
    "data/templeRing/templeR_ang.txt": "0.0 0.0 templeR0001.png\n",
    "data/templeRing_pgm/templeR0001.pgm": "P5\n4 3\n15\n" + "x" * 12,          # maxval 15: the first image read fails
    "bad.json": '{"common": {"klt": {"iters": 4,}}}',
    "nopar/readme.txt": "no par file here\n",
}
CLI_ARGS = [
    [],
    ["data"],
    ["data", "out", "--help"],
    ["data", "out", "-h", "--bogus"],
    ["data", "out", "--bogus"],
    ["data", "out", "5", "--bogus"],
    ["data", "out", "--config"],
    ["data", "out", "--mesh-kf"],
    ["data", "out", "--export-geometry", "cloud"],
    ["data", "out", "--export-geometry", "mesh_stereo", "--mesh-kf", "1", "--mesh-max-points", "10", "--mesh-grid-px", "2", "--mesh-max-edge-px", "3.5"],
    ["data", "out", "--config", "missing.json"],
    ["data", "out", "--config", "bad.json"],
    ["nopar", "out"],
    ["data", "out", "3"],
]


def cli_cases():
    """argument / config / first-read errors of the reference CLI (everything it reports before the per-frame loop)"""
    import subprocess
    cli = H.ref_cli()
    cases = []
    with tempfile.TemporaryDirectory() as td:
        for rel, text in CLI_FILES.items():
            os.makedirs(os.path.dirname(os.path.join(td, rel)), exist_ok=True)
            with open(os.path.join(td, rel), "w") as f:
                f.write(text)
        for args in CLI_ARGS:
            p = subprocess.run([cli] + args, cwd=td, capture_output=True, text=True)
            cases.append(dict(args=args, rc=p.returncode, stdout=p.stdout, stderr=p.stderr.replace(cli, "<CLI>")))
    return dict(files=CLI_FILES, cases=cases)


def main():
    r = H.ref()
    assert r is not None and r.has("ref_read_pgm"), "oracle/_ref/libsfmref.so missing or stale: run `make -C oracle ref`"
    out = {"pgm": [], "json": []}
    with tempfile.TemporaryDirectory() as td:
        for name, data in pgm_inputs():
            path = os.path.join(td, "f.pgm")
            with open(path, "wb") as f:
                f.write(data)
            w, h, cs = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_ulonglong(0)
            err = ctypes.create_string_buffer(512)
            rc = r.dll.ref_read_pgm(path.encode(), ctypes.byref(w), ctypes.byref(h), ctypes.byref(cs), err, 512)
            # a header integer that does not parse leaves the reference's later `int h, maxv` UNINITIALISED (pgm_io.hpp:43-47):
            # which of its two error texts follows is undefined behaviour -- marked so the test accepts either
            ub = name in ("comment_after_newline", "non_numeric_width")
            out["pgm"].append(dict(name=name, ub=ub, data=base64.b64encode(data).decode(), rc=rc, w=w.value if rc == 0 else None,
                                   h=h.value if rc == 0 else None, checksum=cs.value if rc == 0 else None,
                                   error=err.value.decode().replace(path, "<PATH>") if rc else None))
    for name, text, sec, key, kind in json_inputs():
        num = ctypes.c_double(0)
        buf = ctypes.create_string_buffer(512)
        rc = r.dll.ref_config_lookup(text.encode(), sec.encode(), key.encode(), kind, ctypes.byref(num), buf, 512)
        out["json"].append(dict(name=name, text=text, section=sec, key=key, kind=kind, rc=rc, number=num.value if rc == 1 and kind < 2 else None,
                                string=base64.b64encode(buf.value).decode() if (rc == 1 and kind == 2) or rc == -1 else None))
    out["cli"] = cli_cases()
    with open(os.path.join(HERE, "surface_cases.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote surface_cases.json:", len(out["pgm"]), "PGM cases,", len(out["json"]), "JSON cases")
    for c in out["pgm"]:
        print(" ", c["name"], c["rc"], c["error"] or (c["w"], c["h"], c["checksum"]))


if __name__ == "__main__":
    main()
