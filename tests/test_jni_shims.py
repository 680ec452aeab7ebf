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
