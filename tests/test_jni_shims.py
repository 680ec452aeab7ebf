"""The JNI shims a maintainer would add to the reference (integration/jni/*.c) have never met a JDK: the image has none.
What can be checked without one: (1) they parse and type-check against the C ABI headers and a stand-in <jni.h> that declares the
JNIEnv members they use with the JNI specification's signatures (tests/jni_stub/jni.h; -fsyntax-only, nothing is linked);
(2) every `native` method of the Java classes beside them has a shim of the mangled name with the matching C parameter list
(the convention of jni/jgi_BBMergeOverlapper.h:22-39: JNIEnv*, jclass, then one C parameter per Java parameter), and no shim is left over."""
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIMS = sorted(glob.glob(os.path.join(ROOT, "integration", "jni", "*.c")))
JAVA = {"BBDukGpu.c": ("bbduk", os.path.join(ROOT, "integration", "java", "bbduk", "BBDukGpu.java")),
        "SealGpu.c": ("jgi", os.path.join(ROOT, "integration", "java", "jgi", "SealGpu.java"))}
JTYPE = {"int": "jint", "long": "jlong", "boolean": "jboolean", "float": "jfloat", "void": "void", "int[]": "jintArray", "long[]": "jlongArray",
         "float[]": "jfloatArray", "byte[]": "jbyteArray", "ByteBuffer": "jobject", "String": "jstring"}


@pytest.mark.parametrize("src", SHIMS, ids=[os.path.basename(s) for s in SHIMS])
def test_shim_parses_against_the_abi_headers(src):
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-Wno-unused-parameter", "-fsyntax-only",
           "-I" + os.path.join(ROOT, "tests", "jni_stub"), "-I" + os.path.join(ROOT, "include"), src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


def _natives(java_path):
    text = re.sub(r"//[^\n]*", "", open(java_path).read())
    out = {}
    for m in re.finditer(r"native\s+([\w\[\]]+)\s+(\w+)\s*\(([^)]*)\)\s*;", text, re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        ptypes = [p.split()[0] for p in params.split(",") if p.strip()]
        out[name] = (JTYPE[ret], [JTYPE[t] for t in ptypes])
    return out


def _shims(c_path):
    text = re.sub(r"//[^\n]*", "", open(c_path).read())
    out = {}
    for m in re.finditer(r"JNIEXPORT\s+(\w+)\s+JNICALL\s+(Java_\w+)\s*\(([^)]*)\)", text, re.S):
        params = [" ".join(p.split()[:-1]) for p in m.group(3).split(",")]
        out[m.group(2)] = (m.group(1), params)
    return out


@pytest.mark.parametrize("src", SHIMS, ids=[os.path.basename(s) for s in SHIMS])
def test_every_native_method_has_its_shim(src):
    pkg, java_path = JAVA[os.path.basename(src)]
    cls = os.path.splitext(os.path.basename(java_path))[0]
    nat, shims = _natives(java_path), _shims(src)
    assert len(nat) >= 9
    for name, (ret, ptypes) in nat.items():
        sym = "Java_%s_%s_%s" % (pkg, cls, name)
        assert sym in shims, "no shim for native %s" % name
        cret, cparams = shims[sym]
        assert cret == ret, (sym, cret, ret)
        assert cparams[0] == "JNIEnv*" and cparams[1] == "jclass", (sym, cparams[:2])      # static natives
        assert cparams[2:] == ptypes, (sym, cparams[2:], ptypes)
    left = set(shims) - {"Java_%s_%s_%s" % (pkg, cls, n) for n in nat}
    assert not left, "shims without a native declaration: %s" % sorted(left)


def _split_args(text):
    """top-level comma split of a Java argument list"""
    out, depth, cur = [], 0, ""
    for ch in text:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def test_batcher_calls_only_what_bbdukgpu_declares():
    """integration/java/bbduk/BBDukGpuBatcher.java (the pass A / device call / pass B aggregator of INTEGRATION.md section 2) has never met javac either:
    every BBDukGpu.<method>(...) it calls must be a public static method of BBDukGpu.java with that many parameters, every BBDukGpu.<CONSTANT> it names
    must be declared there, and the operators it drives must cover every mode constant."""
    d = os.path.join(ROOT, "integration", "java", "bbduk")
    strip = lambda t: re.sub(r"/\*.*?\*/", "", re.sub(r"//[^\n]*", "", t), flags=re.S)
    gpu, bat = strip(open(os.path.join(d, "BBDukGpu.java")).read()), strip(open(os.path.join(d, "BBDukGpuBatcher.java")).read())
    decl = {}
    for m in re.finditer(r"public\s+static\s+[\w\[\]<>]+\s+(\w+)\s*\(([^)]*)\)\s*\{", gpu, re.S):
        decl[m.group(1)] = [p.split()[0] for p in _split_args(m.group(2))]
    consts = set(re.findall(r"\b([A-Z][A-Z0-9_]+)\s*=\s*\d+", gpu))
    calls = 0
    for m in re.finditer(r"BBDukGpu\.(\w+)\s*\(", bat):
        name, i, depth = m.group(1), m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(bat[i], 0); i += 1
        args = _split_args(bat[m.end():i - 1])
        assert name in decl, "BBDukGpu has no public static %s" % name
        assert len(args) == len(decl[name]), (name, args, decl[name])
        calls += 1
    assert calls >= 8
    used = set(re.findall(r"BBDukGpu\.([A-Z][A-Z0-9_]+)\b", bat))
    assert used <= consts, used - consts
    assert {c for c in consts if c.startswith("MODE_")} - {"MODE_KFILTER"} <= used      # (kfilter is the switch's default branch)
    for t in ("shared.TrimRead", "stream.Read", "structures.ListNum"):                  # the reference classes it leans on, by their real packages
        assert "import %s;" % t in bat
