"""docs/OPTIONS.md, the library and the tests in step (CPU): every option name dpdf_set_option accepts has exactly one row in the table, every
row names an existing option with the default the source has, and the row's owner test mentions the option."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_every_option_has_a_row_a_default_and_an_owner_test():
    csrc = ROOT / "dpdfnet_amd" / "csrc"
    src = (csrc / "host_model_types.h").read_text() + (csrc / "host_create.h").read_text()
    body = src[src.index('extern "C" int dpdf_set_option('):]
    body = body[: body.index("\n}\n")]
    # (names behind DPDF_HAZARD_PROBE exist in probe builds only)
    shipped = re.sub(r"#ifdef DPDF_HAZARD_PROBE.*?#endif", "", body, flags=re.S)
    names = sorted(set(re.findall(r'n == "([a-z0-9_]+)"', shipped)))
    assert len(names) >= 20
    rows = {}
    for line in (ROOT / "docs" / "OPTIONS.md").read_text().splitlines():
        m = re.match(r"\| `([a-z0-9_]+)` \| (-?\d+) \| .* \| `(tests/[a-z_]+\.py)::(test_[a-z0-9_]+)` \|$", line)
        if m:
            assert m.group(1) not in rows, f"{m.group(1)} has two rows"
            rows[m.group(1)] = (int(m.group(2)), m.group(3), m.group(4))
    assert sorted(rows) == names, (sorted(set(names) - set(rows)), sorted(set(rows) - set(names)))
    for name, (dflt, tfile, tname) in rows.items():
        member = "use_gru256_cluster" if name == "gru256_cluster" else name
        m = re.search(r"^\s+(?:int|bool) %s = (-?\d+|true|false);" % member, src, flags=re.M)
        assert m, f"no member with a default for {name}"
        have = {"true": 1, "false": 0}.get(m.group(1), None)
        have = int(m.group(1)) if have is None else have
        assert have == dflt, (name, have, dflt)
        text = (ROOT / tfile).read_text()
        i = text.index(f"def {tname}(")
        j = text.find("\ndef ", i + 1)
        assert f'"{name}"' in text[i: j if j > 0 else len(text)], f"{tfile}::{tname} does not mention {name}"
