#!/usr/bin/env python3
"""Transcribe the reference's golden vectors for the tree-likelihood path into JSON fixtures.

Run in the build container only (reads /root/reference, which does not exist on the GPU box):

    python tests/golden/make_fixtures.py

Outputs (data only — inputs and expected outputs, no reference source text):
  primates.json          6 primate mtDNA sequences x 768 sites, the fixed tree and the PAUP* lnL values asserted by
                         src/test/dr/evomodel/treedatalikelihood/TreeDataLikelihoodTest.java:116-315 and
                         src/test/dr/evomodel/treelikelihood/LikelihoodTest.java:86-345
                         (sequences: src/test/dr/inference/trace/TraceCorrelationAssert.java:192-198,
                          node heights: :145-190)
  branch_specific.json   4 taxa x 14 sites, two GTR models, 1e-13 values from
                         tests/TestXML/testBranchSpecificSubstitutionModel.xml:44-61, 77-79, 114-207, 209-235
  jar_smoke.json         the API-level smoke test baked into lib/beagle.jar (beagle.BeagleFactory#main):
                         3 taxa, literal JC69 eigen system, literal op list, "PAUP logL = -1574.63623"
  transition_probabilities.json   the ten 4 x 4 known-answer matrices of HKYTest / TN93Test / GeneralF81Test (scilab expm, 1e-10)
  benchmark{1,2}_patterns.npz   (python tests/golden/make_fixtures.py --benchmarks) the REAL alignments of
                         examples/Benchmarks/benchmark1.xml (1441 taxa x 987 sites -> 593 patterns, HKY) and benchmark2.xml
                         (62 x 10869 -> 5565, GTR+G4) as unique site patterns + weights; no tree (the XMLs draw a random
                         one) and no expected value (none is published)
"""
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def primate_sequences():
    src = open(os.path.join(REF, "src/test/dr/inference/trace/TraceCorrelationAssert.java")).read()
    block = src[src.index("PRIMATES_TAXON_SEQUENCE"):src.index("DENGUE4_TAXON_SEQUENCE")]
    strings = re.findall(r'"([^"]*)"', block)
    names = strings[:6]
    seqs = strings[6:12]
    assert names == ["human", "chimp", "bonobo", "gorilla", "orangutan", "siamang"], names
    assert all(len(s) == 768 for s in seqs), [len(s) for s in seqs]
    return names, seqs


def java_numbers(text):
    return [float(x) for x in re.findall(r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?", text)]


def transition_probabilities():
    """The known-answer transition-probability matrices of the reference's substitution-model tests (expm of the normalised
    generator, computed with scilab by the tests' authors, asserted to 1e-10): instance by instance the frequencies, the
    model's parameters, the distance and the 16 expected entries.  The relative rates are written out in the row-major
    upper-triangle order A-C, A-G, A-T, C-G, C-T, G-T the tests' own generator comments use
    (XQ = [0 1 k1 1; 1 0 1 k2; k1 1 0 1; 1 k2 1 0], Q = XQ diag(pi) with the diagonal filled in, normalised to one expected
    substitution per unit time)."""
    cases = []
    for path, params in (("src/test/dr/evomodel/substmodel/HKYTest.java", ("getKappa",)),
                         ("src/test/dr/evomodel/substmodel/TN93Test.java", ("getKappa1", "getKappa2")),
                         ("src/test/dr/evomodel/substmodel/GeneralF81Test.java", ())):
        src = open(os.path.join(REF, path)).read()
        src = re.sub(r"//[^\n]*", "", src)
        blocks = re.split(r"Instance\s+(test\d+)\s*=\s*new\s+Instance\s*\(\s*\)", src)
        # blocks = [head, name, body, name, body, ...]; the last body runs on into the test method: cut it at "Instance[] all"
        for name, body in zip(blocks[1::2], blocks[2::2]):
            body = body.split("Instance[]")[0]

            def method(m):
                return re.search(m + r"\s*\(\s*\)\s*\{(.*?)\}\s*;?\s*\}", body, re.S).group(1) if (m + "(") in body.replace(" ", "") else None
            pi = java_numbers(re.search(r"getPi\s*\(\s*\)\s*\{\s*return\s+new\s+double\[\]\s*\{(.*?)\}", body, re.S).group(1))
            expected = java_numbers(re.search(r"getExpectedResult\s*\(\s*\)\s*\{\s*return\s+new\s+double\[\]\s*\{(.*?)\}", body, re.S).group(1))
            distance = java_numbers(re.search(r"getDistance\s*\(\s*\)\s*\{\s*return(.*?);", body, re.S).group(1))[0]
            k = [java_numbers(re.search(m + r"\s*\(\s*\)\s*\{\s*return(.*?);", body, re.S).group(1))[0] for m in params]
            assert len(pi) == 4 and len(expected) == 16 and abs(sum(pi) - 1.0) < 1e-12, (path, name, pi, expected)
            k1, k2 = (k[0], k[0]) if len(k) == 1 else (k[0], k[1]) if len(k) == 2 else (1.0, 1.0)
            cases.append({"source": "%s %s" % (path, name), "model": os.path.basename(path)[:-len("Test.java")], "pi": pi,
                          "relative_rates_ac_ag_at_cg_ct_gt": [1.0, k1, 1.0, 1.0, k2, 1.0], "distance": distance,
                          "expected_row_major": expected, "tolerance": 1e-10})
    assert len(cases) == 10, len(cases)
    return {"what": "known-answer transition-probability matrices asserted by the reference's substitution-model tests "
                    "(HKYTest.java:151-156, TN93Test.java:167-172, GeneralF81Test.java:158-163: |P - expected| <= 1e-10 entry by entry)",
            "cases": cases}


def benchmark_alignment(xml_name):
    """Taxon names and sequences of an examples/Benchmarks XML (data only)."""
    import xml.etree.ElementTree as ET
    root = ET.parse(os.path.join(REF, "examples", "Benchmarks", xml_name)).getroot()
    names = [t.get("id") for t in root.find("taxa").findall("taxon")]
    seqs = []
    for s in root.find("alignment").findall("sequence"):
        ref = s.find("taxon").get("idref")
        text = "".join(t for t in s.itertext()).replace(ref, "")
        seqs.append((ref, re.sub(r"\s+", "", text)))
    assert [r for r, _ in seqs] == names, "alignment order differs from the taxa block"
    return names, [q for _, q in seqs]


def benchmark_fixture(xml_name, out_name, expect_taxa, expect_sites, expect_patterns, model, cite):
    """Real alignments of the reference's own benchmark inputs, compressed to unique site patterns exactly as
    SitePatterns does (state codes of Nucleotides.java incl. the IUPAC ambiguity codes; first occurrence keeps the column).
    The XMLs draw a RANDOM coalescent starting tree and publish no expected lnL, so these fixtures pin engine <-> oracle on
    real ambiguity patterns, pattern weights and sizes — not an absolute value."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from beast_mcmc_amd.inputs import patterns
    names, seqs = benchmark_alignment(xml_name)
    assert len(names) == expect_taxa and all(len(q) == expect_sites for q in seqs), (len(names), set(len(q) for q in seqs))
    rows = np.stack([patterns.nucleotide_states(q) for q in seqs])
    pats, weights = patterns.site_patterns(rows, unique=True)
    assert pats.shape[1] == expect_patterns, pats.shape            # the XML's own "npatterns=" comment
    np.savez_compressed(os.path.join(HERE, out_name), patterns=pats.astype(np.uint8), weights=weights,
                        taxa=np.array(names), model=json.dumps(model), source=cite)
    print("wrote %s: %d taxa x %d sites -> %d unique patterns (%d ambiguous cells)"
          % (out_name, len(names), expect_sites, pats.shape[1], int((pats > 3).sum())))


def main():
    if "--benchmarks" in __import__("sys").argv:
        benchmark_fixture("benchmark1.xml", "benchmark1_patterns.npz", 1441, 987, 593,
                          {"model": "hky", "kappa": 2.0, "pi": [0.25] * 4, "gamma_categories": 1, "clock_rate": 3.6e-4,
                           "popSize_years": 52.0},
                          "examples/Benchmarks/benchmark1.xml:11 (taxa), :4340-10106 (alignment), :10108-10109 (npatterns=593), "
                          ":10161-10186 (HKY kappa 2, equal frequencies, no gamma), :10154-10158 (clock 3.6E-4)")
        benchmark_fixture("benchmark2.xml", "benchmark2_patterns.npz", 62, 10869, 5565,
                          {"model": "gtr", "rates": [1.0, 1.0, 1.0, 1.0, 1.0, 1.0], "pi": [0.25] * 4, "gamma_categories": 4, "alpha": 0.5,
                           "root_height": 0.2},
                          "examples/Benchmarks/benchmark2.xml:11 (taxa), :79-638 (alignment), :641 (npatterns=5565), :702-735 (GTR, "
                          "alpha 0.5, 4 categories), :664 (rootHeight 0.2)")
        return
    names, seqs = primate_sequences()
    primates = {
        "source": "src/test/dr/inference/trace/TraceCorrelationAssert.java:145-198",
        "taxa": names,
        "sequences": seqs,
        # createPrimateTreeModel(): ((((human,(chimp,bonobo)),gorilla),orangutan),siamang), node heights
        "tree_nested": [[[[0, [1, 2, 0.010772], 0.024003], 3, 0.036038], 4, 0.069125], 5, 0.099582],
        "newick": "((((human:0.024003,(chimp:0.010772,bonobo:0.010772):0.013231):0.012035,gorilla:0.036038):0.033087,"
                  "orangutan:0.069125):0.030457,siamang:0.099582);",
        # model: "hky"/"gtr"; pi: "equal"/"empirical"; alpha/cats: gamma; pinv: invariant class
        "tree_data_likelihood_test": [
            {"name": "JC69", "model": "hky", "kappa": 1.0, "pi": "equal", "lnL": -1992.20564, "cite": "TreeDataLikelihoodTest.java:116-132"},
            {"name": "K80", "model": "hky", "kappa": 8.0, "pi": "equal", "lnL": -1868.89782, "cite": ":135-147"},
            {"name": "HKY85", "model": "hky", "kappa": 8.0, "pi": "empirical", "lnL": -1839.84514, "cite": ":149-161"},
            {"name": "HKY85G", "model": "hky", "kappa": 8.0, "pi": "empirical", "alpha": 0.5, "cats": 4, "lnL": -1816.82611, "cite": ":163-178"},
            {"name": "HKY85I", "model": "hky", "kappa": 8.0, "pi": "empirical", "pinv": 0.75, "lnL": -1822.37478, "cite": ":180-196"},
            {"name": "HKY85GI", "model": "hky", "kappa": 8.0, "pi": "empirical", "alpha": 0.5, "cats": 4, "pinv": 0.75, "lnL": -1815.02176, "cite": ":198-214"},
            {"name": "GTR", "model": "gtr", "rates": [1, 1, 1, 1, 1, 1], "pi": "empirical", "lnL": -1969.14584, "cite": ":216-238"},
            {"name": "GTRI", "model": "gtr", "rates": [1, 1, 1, 1, 1, 1], "pi": "empirical", "pinv": 0.5, "lnL": -1948.84175, "cite": ":240-262"},
            {"name": "GTRG", "model": "gtr", "rates": [1, 1, 1, 1, 1, 1], "pi": "empirical", "alpha": 0.5, "cats": 4, "lnL": -1949.03601, "cite": ":264-286"},
            {"name": "GTRGI", "model": "gtr", "rates": [1, 1, 1, 1, 1, 1], "pi": "empirical", "alpha": 0.5, "cats": 4, "pinv": 0.5, "lnL": -1951.62188, "cite": ":288-315"},
        ],
        # old TreeLikelihood path (LikelihoodTest.java): same pruning arithmetic, PAUP-optimised parameters,
        # site rates from the OLDER discretisation src/dr/oldevomodel/sitemodel/GammaSiteModel.java:271-311
        # (textbook Gamma+I mixture) — hence the different GTR+G+I value.
        "likelihood_test": [
            {"name": "JC69", "model": "hky", "kappa": 1.0, "pi": "equal", "lnL": -1992.20564, "cite": "LikelihoodTest.java:86-107"},
            {"name": "K80", "model": "hky", "kappa": 27.402591, "pi": "equal", "lnL": -1856.30305, "cite": ":109-129"},
            {"name": "HKY85", "model": "hky", "kappa": 29.739445, "pi": "empirical", "lnL": -1825.21317, "cite": ":132-152"},
            {"name": "HKY85G", "model": "hky", "kappa": 38.829740, "pi": "empirical", "alpha": 0.137064, "cats": 4, "lnL": -1789.75936, "cite": ":155-176"},
            {"name": "HKY85I", "model": "hky", "kappa": 38.564672, "pi": "empirical", "pinv": 0.701211, "lnL": -1789.91240, "cite": ":179-200"},
            {"name": "HKY85GI", "model": "hky", "kappa": 39.464538, "pi": "empirical", "alpha": 0.587649, "cats": 4, "pinv": 0.486548, "lnL": -1789.63923, "cite": ":203-225"},
            {"name": "GTR", "model": "gtr", "rates": [1, 1, 1, 1, 1, 1], "pi": "empirical", "lnL": -1969.14584, "cite": ":228-253"},
            {"name": "GTRI", "model": "gtr", "rates": [1, 1, 1, 1, 1, 1], "pi": "empirical", "pinv": 0.5, "lnL": -1948.84175, "cite": ":256-282"},
            {"name": "GTRG", "model": "gtr", "rates": [1, 1, 1, 1, 1, 1], "pi": "empirical", "alpha": 0.5, "cats": 4, "lnL": -1949.03601, "cite": ":285-311"},
            {"name": "GTRGI", "model": "gtr", "rates": [1, 1, 1, 1, 1, 1], "pi": "empirical", "alpha": 0.5, "cats": 4, "pinv": 0.5, "lnL": -1947.58294, "cite": ":314-341"},
        ],
    }
    json.dump(primates, open(os.path.join(HERE, "primates.json"), "w"), indent=1)

    branch = {
        "source": "tests/TestXML/testBranchSpecificSubstitutionModel.xml:44-61,77-79,114-207,209-235",
        "taxa": ["A", "B", "C", "D"],
        "sequences": ["AAACCCGGTAACAA", "AAACCTGGGAATAA", "AAACTCGGGAATGA", "ATACCCGGTGGTAG"],
        "tree_nested": [0, [1, [2, 3, 1.0], 2.0], 3.0],        # (A:1,(B:1,(C:1,D:1):1):1): internal heights 1, 2, 3
        "tip_heights": [2.0, 1.0, 0.0, 0.0],                   # every branch has length 1
        "clock_rate": 0.1,
        "gtr1": {"pi": [0.1, 0.2, 0.3, 0.4], "rates": [1.0, 2.0, 1.0, 1.0, 2.0, 1.0]},
        "gtr2": {"pi": [0.4, 0.3, 0.2, 0.1], "rates": [1.0, 25.0, 1.0, 1.0, 25.0, 1.0]},
        "alpha": 0.5, "cats": 4,
        "clade": ["C", "D"],
        "tolerance": 1e-13,
        "cases": [{"stem_weight": 0.0, "lnL": -68.07217469138813}, {"stem_weight": 1.0, "lnL": -67.91898348796958}],
    }
    json.dump(branch, open(os.path.join(HERE, "branch_specific.json"), "w"), indent=1)

    smoke = {
        "source": "lib/beagle.jar!beagle/BeagleFactory.class#main (literal arrays and string constants); "
                  "sequences as src/test/dr/app/beagle/TinyTest.java:101-107",
        "taxa": ["human", "chimp", "gorilla"],
        "sequences": [seqs[0], seqs[1], seqs[3]],
        "instance": {"tipCount": 3, "partialsBufferCount": 10, "compactBufferCount": 3, "stateCount": 4,
                     "eigenBufferCount": 1, "matrixBufferCount": 4, "categoryCount": 1, "scaleBufferCount": 3},
        "evec": [1.0, 2.0, 0.0, 0.5, 1.0, -2.0, 0.5, 0.0, 1.0, 2.0, 0.0, -0.5, 1.0, -2.0, -0.5, 0.0],
        "ivec": [0.25, 0.25, 0.25, 0.25, 0.125, -0.125, 0.125, -0.125, 0.0, 1.0, 0.0, -1.0, 1.0, 0.0, -1.0, 0.0],
        "eval": [0.0, -1.3333333333333333, -1.3333333333333333, -1.3333333333333333],
        "freqs": [0.25, 0.25, 0.25, 0.25],
        "rates": [1.0], "weights": [1.0],
        "matrix_indices": [0, 1, 2, 3],
        "edge_lengths": [0.1, 0.1, 0.2, 0.1],
        # {dest, writeScale, readScale, child1, matrix1, child2, matrix2}; the jar passes 0/1 in the scale
        # fields, which its Java implementation ignores -> NONE (-1) here
        "operations": [3, -1, -1, 0, 0, 1, 1, 4, -1, -1, 2, 2, 3, 3],
        "root": 4,
        "lnL": -1574.63623,
        "decimals": 5,
    }
    json.dump(smoke, open(os.path.join(HERE, "jar_smoke.json"), "w"), indent=1)

    epoch = {
        "source": "tests/TestXML/testEpochConvolutionOrder.xml:9-19 (taxa, dates, states), :31-33 (tree (X:7,Y:3)), :46-49 (clock 0.1), "
                  ":60-119 (four asymmetric 2-state models), :120-132 (epoch transition times 2, 4, 6), :179-193 (expected value)",
        "taxa": ["X", "Y"], "tip_heights": [0.0, 4.0], "root_height": 7.0, "states": [0, 1],
        "clock_rate": 0.1,
        "root_freqs": [0.5, 0.5],
        # generalSubstitutionModel with S(S-1) = 2 relative rates {0->1, 1->0}, frequencies 0.5/0.5, normalised
        "epoch_rates": [[0.9, 0.1], [0.7, 0.3], [0.2, 0.8], [0.5, 0.5]],
        "transition_times": [2.0, 4.0, 6.0],
        "lnL": -2.17228, "tolerance": 1e-3,
        "note": "branch matrix = product of the per-epoch matrices from the ROOT end of the branch to the tip end "
                "(the test's name: the other order gives -1.74802)",
    }
    json.dump(epoch, open(os.path.join(HERE, "epoch_convolution.json"), "w"), indent=1)
    json.dump(transition_probabilities(), open(os.path.join(HERE, "transition_probabilities.json"), "w"), indent=1)
    print("wrote primates.json, branch_specific.json, jar_smoke.json, epoch_convolution.json, transition_probabilities.json")


if __name__ == "__main__":
    main()
