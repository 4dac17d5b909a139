"""Every file under profiles/, tests/, tools/, oracle/, include/, plugin/ that DESIGN.md, README.md or INTEGRATION.md
names must exist in the tree (a figure whose record was not kept gets no credit -- and should not be cited)."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
PAT = re.compile(r"`((?:profiles|tests|tools|oracle|include|plugin)/[A-Za-z0-9_./*-]+)`")


def test_cited_files_exist():
    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md"):
        for m in PAT.finditer((ROOT / doc).read_text()):
            p = m.group(1).rstrip(".")
            if p.startswith("oracle/_ref"):
                continue                                # built, git-ignored
            if "*" in p:
                if not list(ROOT.glob(p)):
                    missing.append((doc, p))
            elif not (ROOT / p).exists() and not list(ROOT.glob(p + "*")):      # `profiles/r03_b_batch64` = a prefix of two files
                missing.append((doc, p))
    assert not missing, missing
