"""Planning / MPC call surface (mirror of the reference's `confrez/control`)."""
