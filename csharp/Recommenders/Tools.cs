// Behaviour of the reference's Recommenders/Tools.cs:6-9 (Program.cs:72 calls it): prints the stopwatch's
// elapsed whole milliseconds as a TimeSpan.  Not on the hot path; kept so the host compiles unchanged.
using System;
using System.Diagnostics;

namespace Recommenders {
    public class Tools {
        public static void printExecutionTime(Stopwatch stopwatch) {
            long elapsedMs = stopwatch.ElapsedMilliseconds;
            Console.WriteLine("Execution time: " + TimeSpan.FromMilliseconds(elapsedMs));
        }
    }
}
