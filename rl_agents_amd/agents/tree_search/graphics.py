"""Pictures of an exported search tree (reference ``rl_agents/agents/tree_search/graphics.py``).

``TreePlot`` is the matplotlib figure ``AbstractTreeSearchAgent.write_tree`` sends to the summary writer when
``display_tree`` is set (abstract.py:104-106, graphics.py:115-166): one line segment per visited child, fanned out by
action index, its width growing with the child's share of the root's visits.  It reads only what the exported
:class:`~rl_agents_amd.agents.tree_search.abstract.Node` tree offers -- ``planner.root``, ``children``, ``count`` -- so the
reference's own ``TreePlot`` / ``TreeGraphics`` work on the same object (tests/test_tree_tools.py feeds it one).
matplotlib is optional: without it ``write_tree`` logs once and does nothing.
"""
import logging

import numpy as np

logger = logging.getLogger(__name__)


def _pyplot():
    try:
        import matplotlib
        import matplotlib.pyplot as plt
        return matplotlib, plt
    except Exception:                                   # pragma: no cover - depends on the image
        return None, None


class TreePlot(object):
    def __init__(self, planner, max_depth=4):
        self.planner = planner
        self.actions = int(planner.env.action_space.n)
        self.max_depth = int(max_depth)
        root = planner.root
        self.total_count = sum(c.count for c in root.children.values()) if root is not None else 0

    def line_width(self, count):
        """0.5 pt for a single visit up to 4 pt for every visit of the root (graphics.py:143 with utils.remap/constrain)."""
        if self.total_count <= 1:
            return 0.5
        w = 0.5 + (count - 1) * (4 - 0.5) / float(self.total_count - 1)
        return float(min(max(w, 0.5), 4.0))

    def segments(self):
        """[(x0, y0, x1, y1, width)] of every drawn edge: children with at least one visit, down to ``max_depth``."""
        out = []
        root = self.planner.root
        if root is None:
            return out
        stack = [(root, 0.0, 0.0, 0)]
        while stack:
            node, x, y, depth = stack.pop()
            if depth > self.max_depth:
                continue
            spread = 1.0 / self.actions ** depth
            for a in range(self.actions):
                child = node.children.get(a)
                if child is None or not child.count:
                    continue
                cx = x - spread / 2 + (a / (self.actions - 1) * spread if self.actions > 1 else spread / 2)
                cy = y - 1.0 / self.max_depth
                out.append((x, y, cx, cy, self.line_width(child.count)))
                stack.append((child, cx, cy, depth + 1))
        return out

    def plot(self, filename, title=None, ax=None):
        _, plt = _pyplot()
        if plt is None:
            logger.warning("matplotlib is not available: the tree is not plotted")
            return None
        if ax is None:
            _, ax = plt.subplots()
        for x0, y0, x1, y1, w in self.segments():
            ax.plot([x0, x1], [y0, y1], "k", linewidth=w, solid_capstyle="round")
        if title:
            ax.set_title(title)
        ax.axis("off")
        if filename is not None:
            import os
            os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
            ax.figure.savefig(filename, dpi=300)
        return ax

    def plot_to_writer(self, writer, epoch=0, figsize=None, show=False):
        """The plot as a [3, H, W] uint8 image to ``writer.add_image("Expanded_tree", image, epoch)``."""
        matplotlib, plt = _pyplot()
        if plt is None:
            logger.warning("matplotlib is not available: display_tree is ignored")
            return None
        fig = plt.figure(figsize=figsize, tight_layout=True)
        ax = fig.add_subplot(111)
        title = "Expanded_tree"
        self.plot(filename=None, title=title, ax=ax)
        fig.canvas.draw()
        rgba = np.asarray(fig.canvas.buffer_rgba())
        image = np.ascontiguousarray(np.moveaxis(rgba[..., :3], 2, 0))
        if writer:
            writer.add_image(title, image, epoch)
        if show and matplotlib.get_backend().lower() not in ("agg", "pdf", "svg", "ps", "cairo", "template"):
            plt.show()                                  # (a blocking window only on an interactive backend)
        plt.close(fig)
        return image
