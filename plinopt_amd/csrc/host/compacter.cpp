// ===========================================================================
// bin/compacter -- post-pass on the straight-line program of the single winner
// (reference src/compacter.cpp:27-68 `Compacter`, include/plinopt_programs.inl:
// 1157-1455 `variablesTrimer`; every reference pipeline reads
// `optimizer | compacter -s | SLPchecker`, bin/FDT.sh:58, Makefile:79-80, and
// bin/GDT.sh:41-65 pins the operation count of its output on the stored programs).
// Prints the reference's text: the engine (plo_trim.hpp) follows variablesTrimer pass by
// pass and is held line by line to the literal restatement oracle/plo_compact_oracle.py.
// Statistics on stderr as the reference ("# N elements instead of M" per round).
// Host only: it runs once, on one program.
// Usage: compacter [-s|-n] [-O #] [stdin|file.slp]
// ===========================================================================
#include "plo_trim.hpp"
#include <fstream>
#include <iostream>

int main(int argc, char **argv)
{
    bool simplSingle = true; std::string filename; size_t numloops = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a(argv[i]);
        if (a == "-h") {
            std::clog << "Usage: " << argv[0] << " [-s/-n] [-O #] [stdin|file.prg]\n"
                      << "  -s/-n: replace/not-replace singly used variables\n"
                      << "  -O #: number of trim loops (default until stable)\n";
            exit(-1);
        } else if (a == "-s") simplSingle = true;
        else if (a == "-n" || a == "-ns") simplSingle = false;
        else if (a == "-O" && i + 1 < argc) numloops = strtoull(argv[++i], nullptr, 10);
        else filename = a;
    }
    try {
        plo::trim::Engine E;
        plo::trim::Prog P;
        if (filename.empty()) P = plo::trim::parse(std::cin, E.W);
        else { std::ifstream in(filename); if (!in) return 0; P = plo::trim::parse(in, E.W); }
        const size_t PVs = plo::trim::elements(P);
        std::clog << std::string(40, '#') << std::endl;
        E.run(P, simplSingle, numloops, &std::clog);
        plo::trim::print(std::cout, P, E.W);
        std::clog << "# \033[1;32m" << plo::trim::elements(P) << "\telements\tinstead of " << PVs << "\033[0m" << std::endl;
        std::clog << std::string(40, '#') << std::endl;
    } catch (const std::exception &e) {
        std::cerr << "# \033[1;31mERROR: " << e.what() << "\033[0m" << std::endl;
        return 3;
    }
    return 0;
}
