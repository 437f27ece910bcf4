"""tests/cpp/refstub restates DECLARATIONS of the reference (so that the reference-side binding can be compile-checked in
an image without Eigen / iDynTree / BLF).  Where the reference is readable (this container; never the GPU box) every
member the stand-ins declare is looked up in the reference's own header text: same return type, same parameter list,
same cv-qualifier.  And the code block of INTEGRATION.md section 2 must be tests/cpp/integration_snippet.cpp, verbatim."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "cpp", "refstub")
REF_UTILS = "/root/reference/src/flight-controller/utils/include"


def _norm(text):
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"\s+", " ", text)


def _public_member_declarations(stub_text):
    """signatures of the stand-in's public members that mirror the reference (everything before the first test* loader)"""
    body = stub_text[stub_text.index("public:") + len("public:"):]
    cut = [body.find(m) for m in ("// ---- test", "private:") if body.find(m) >= 0]
    body = _norm(body[:min(cut)])
    # <return type> name(<params>) [const]   followed by an inline body
    pat = re.compile(r"((?:const )?[\w:<>, ]+?[&\*]?) (\w+)\(([^()]*)\)( const)? \{")
    out = []
    for m in pat.finditer(body):
        ret, name, params, cv = m.group(1).strip(), m.group(2), m.group(3).strip(), (m.group(4) or "").strip()
        if name in ("if", "for", "while", "switch"):
            continue
        out.append((ret, name, params, cv))
    return out


@pytest.mark.skipif(not os.path.isdir(REF_UTILS), reason="the reference is only readable in the build container")
@pytest.mark.parametrize("header, expect_at_least", [("QPInput.h", 18), ("Robot.h", 19), ("TrajectoryManager.h", 5)])
def test_stub_declarations_are_the_references(header, expect_at_least):
    ref = _norm(open(os.path.join(REF_UTILS, header)).read())
    decls = _public_member_declarations(open(os.path.join(STUB, header)).read())
    assert len(decls) >= expect_at_least, decls
    missing = []
    for ret, name, params, cv in decls:
        sig = f"{ret} {name}({params}){' ' + cv if cv else ''};"
        # the reference breaks lines inside declarations: compare with all whitespace removed
        if sig.replace(" ", "") not in ref.replace(" ", ""):
            missing.append(sig)
    assert not missing, f"{header}: not declared like this in the reference: {missing}"


def test_integration_md_shows_the_compiled_snippet_verbatim():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    snippet = open(os.path.join(ROOT, "tests", "cpp", "integration_snippet.cpp")).read().strip()
    blocks = re.findall(r"```cpp\n(.*?)```", doc, flags=re.S)
    assert any(b.strip() == snippet for b in blocks), "INTEGRATION.md section 2 must carry tests/cpp/integration_snippet.cpp verbatim"
